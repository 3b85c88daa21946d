"""FastDPM accelerated sampling (S << T network calls).

Mirrors reference pointnet2/util_fastdpmv2.py: bisearch (:186-209), get_VAR_noise
(:212-236), get_STEP_step (:239-258), _log_gamma / _log_cont_noise (:261-272),
_precompute_VAR_steps (:275-304), VAR_sampling (:307-381), STEP_sampling (:384-452),
fast_sampling_function_v2 (:455-476).

The host-side schedule search runs in float64 EXPLICITLY.  The reference passes
0-d float32 NumPy arrays into _log_cont_noise (:297-299); under NumPy 1.x those
promoted to float64, under NumPy >= 2 they stay float32 and the last continuous
step comes out 0.497, tripping the reference's own `assert abs(tau) < 0.1`.  The
float64 behaviour is the one the published S=50 schedules correspond to.
"""
import numpy as np
import torch

from .util import _expand_label, get_device, std_normal


def bisearch(f, domain, target, eps=1e-8):
    """Bisection for f decreasing on `domain`: returns x with f(x) within (1 +- eps) of target."""
    sign = -1 if target < 0 else 1
    left, right = domain
    x = (left + right) / 2
    for _ in range(1000):
        x = (left + right) / 2
        fx = f(x)
        if fx < target:
            right = x
        elif fx > (1 + sign * eps) * target:
            left = x
        else:
            break
    return x


def get_VAR_noise(S, diffusion_config, schedule='linear'):
    """S noise levels eta with prod(1-eta) = alpha_bar_T of the full T-step schedule."""
    b0, bT, T = diffusion_config["beta_0"], diffusion_config["beta_T"], diffusion_config["T"]
    target = np.prod(1 - np.linspace(b0, bT, T))
    if schedule == 'linear':
        g = lambda x: np.linspace(b0, x, S)
        domain = (b0, 0.99)
    elif schedule == 'quadratic':
        g = lambda x: np.array([b0 * (1 + i * x) ** 2 for i in range(S)])
        domain = (0.0, 0.95 / np.sqrt(b0) / S)
    else:
        raise NotImplementedError
    return g(bisearch(lambda x: np.prod(1 - g(x)), domain, target, eps=1e-4))


def get_STEP_step(S, diffusion_config, schedule='linear'):
    """S integer steps out of T (linear: evenly spaced; quadratic: dense near 0)."""
    T = diffusion_config["T"]
    if schedule == 'linear':
        c = (T - 1.0) / (S - 1.0)
        taus = [np.floor(i * c) for i in range(S)]
    elif schedule == 'quadratic':
        taus = np.linspace(0, np.sqrt(T * 0.8), S) ** 2
    else:
        raise NotImplementedError
    return [int(s) for s in taus]


def _log_gamma(x):
    # Stirling: Gamma(y+1) ~ sqrt(2 pi y) (y/e)^y (1 + 1/(12 y))
    y = x - 1
    return np.log(2 * np.pi * y) / 2 + y * (np.log(y) - 1) + np.log(1 + 1 / (12 * y))


def _log_cont_noise(t, beta_0, beta_T, T):
    """log alpha_bar at a CONTINUOUS step t of the linear schedule.

    beta_0 / beta_T arrive as float32 values (Beta[0], Beta[-1]).  NumPy-1 scalar
    promotion, which the reference was written against: the difference of the two
    float32 scalars is taken in float32, everything after that is float64."""
    diff32 = np.float32(beta_T) - np.float32(beta_0)
    delta = np.float64(diff32) / (T - 1)
    c = (1.0 - np.float64(np.float32(beta_0))) / delta
    t1 = np.float64(t) + 1
    return t1 * np.log(delta) + _log_gamma(c + 1) - _log_gamma(c - t1 + 1)


def _gamma_bar(user_defined_eta):
    beta_tilde = torch.from_numpy(np.asarray(user_defined_eta)).to(torch.float32)
    g = 1 - beta_tilde
    for t in range(1, len(g)):
        g[t] *= g[t - 1]
    return g


def _precompute_VAR_steps(diffusion_hyperparams, user_defined_eta):
    """For each of the S noise levels find the fractional step tau with alpha_bar(tau) = gamma_bar."""
    dh = diffusion_hyperparams
    T, Alpha_bar, Beta = dh["T"], dh["Alpha_bar"].cpu(), dh["Beta"].cpu()
    assert len(Alpha_bar) == T
    Gamma_bar = _gamma_bar(user_defined_eta)
    assert Gamma_bar[0] <= Alpha_bar[0] and Gamma_bar[-1] >= Alpha_bar[-1]
    b0, bT = Beta[0].numpy(), Beta[-1].numpy()   # 0-d float32, as in the reference
    steps = []
    for t in range(len(Gamma_bar) - 1, -1, -1):
        tau = None
        for i in range(T - 1):
            if Alpha_bar[i] >= Gamma_bar[t] > Alpha_bar[i + 1]:
                tau = bisearch(lambda _t: _log_cont_noise(_t, b0, bT, T), domain=(i - 0.01, i + 1.01),
                               target=float(np.log(Gamma_bar[t].numpy())))   # float32 log, as in the reference
                break
        steps.append(T - 1 if tau is None else tau)
    return steps


def _ddim_update(x, eps, a_cur, a_next, sigma, size):
    x = x * torch.sqrt(a_next / a_cur)
    c = torch.sqrt(1 - a_next - sigma ** 2) - torch.sqrt(1 - a_cur) * torch.sqrt(a_next / a_cur)
    return x + (c * eps + sigma * std_normal(size))   # reference: x += c*eps + sigma*z


def _run(net, size, steps, alpha_of, kappa, label, verbose, condition, last_check):
    dev = get_device()
    x = std_normal(size)
    label = _expand_label(label, size[0])
    n = len(steps)
    with torch.no_grad():
        for i, tau in enumerate(steps):
            if verbose:
                print('t %.2f x max %.2f min %.2f' % (tau, x.max(), x.min()))
            ts = (tau * torch.ones((size[0],))).to(dev)
            if condition is None:
                eps = net(x, ts=ts, label=label)
            else:
                eps = net(x, condition, ts=ts, label=label, use_retained_condition_feature=True)
            if verbose:
                print('t %.2f epsilon_theta max %.2f min %.2f' % (tau, eps.max(), eps.min()))
            a_cur = alpha_of(i)
            if i == n - 1:
                last_check(tau)
                a_next, sigma = torch.tensor(1.0), torch.tensor(0.0)
            else:
                a_next = alpha_of(i + 1)
                sigma = kappa * torch.sqrt((1 - a_next) / (1 - a_cur) * (1 - a_cur / a_next))
            x = _ddim_update(x, eps, a_cur, a_next, sigma, size)
    if condition is not None:
        net.reset_cond_features()
    return x


def VAR_sampling(net, size, diffusion_hyperparams, user_defined_eta, kappa, continuous_steps,
                 print_every_n_steps=100, label=0, verbose=True, condition=None):
    dh = diffusion_hyperparams
    T, Alpha_bar = dh["T"], dh["Alpha_bar"]
    assert len(dh["Alpha"]) == T and len(Alpha_bar) == T and len(dh["Sigma"]) == T
    assert len(size) == 3 and 0.0 <= kappa <= 1.0
    Gamma_bar = _gamma_bar(user_defined_eta)
    S = len(Gamma_bar)
    assert Gamma_bar[0] <= Alpha_bar[0].cpu() and Gamma_bar[-1] >= Alpha_bar[-1].cpu()
    print('begin sampling, total number of reverse steps = %s' % S)

    def last_check(tau):
        assert abs(tau) < 0.1

    return _run(net, size, continuous_steps, lambda i: Gamma_bar[S - 1 - i], kappa, label, verbose, condition,
                last_check)


def STEP_sampling(net, size, diffusion_hyperparams, user_defined_steps, kappa, print_every_n_steps=100, label=0,
                  verbose=True, condition=None):
    dh = diffusion_hyperparams
    T, Alpha_bar = dh["T"], dh["Alpha_bar"]
    assert len(dh["Alpha"]) == T and len(Alpha_bar) == T and len(dh["Sigma"]) == T
    assert len(size) == 3 and 0.0 <= kappa <= 1.0
    steps = sorted(list(user_defined_steps), reverse=True)
    print('begin sampling, total number of reverse steps = %s' % len(steps))
    abar = Alpha_bar.cpu()

    def last_check(tau):
        assert tau == 0

    return _run(net, size, steps, lambda i: abar[steps[i]], kappa, label, verbose, condition, last_check)


def fast_sampling_function_v2(net, size, diffusion_hyperparams, diffusion_config, length=100, sampling_method='var',
                              schedule='quadratic', kappa=0.0, print_every_n_steps=100, label=0, verbose=True,
                              condition=None):
    assert sampling_method in ['var', 'step']
    assert schedule in ['quadratic', 'linear']
    if sampling_method == 'var':
        eta = get_VAR_noise(length, diffusion_config, schedule)
        steps = _precompute_VAR_steps(diffusion_hyperparams, eta)
        return VAR_sampling(net, size, diffusion_hyperparams, eta, kappa, steps,
                            print_every_n_steps=print_every_n_steps, label=label, verbose=verbose,
                            condition=condition)
    steps = get_STEP_step(length, diffusion_config, schedule)
    return STEP_sampling(net, size, diffusion_hyperparams, steps, kappa, print_every_n_steps=print_every_n_steps,
                         label=label, verbose=verbose, condition=condition)
