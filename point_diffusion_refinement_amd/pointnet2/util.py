"""DDPM hyper-parameters and the reverse (ancestral) sampling loop.

Mirrors reference pointnet2/util.py: std_normal (:118-123),
calc_diffusion_hyperparams (:154-181), sampling (:184-255).  One iteration of the
loop body in `sampling` is the "p_sample" step BASELINE.json's north star names.

Noise source: the reference draws every normal on the CPU default generator and
copies it to the GPU (x_T, then z_{T-1} ... z_1), so "identical seeds" means that
CPU stream.  `noise='cpu'` reproduces it; `noise='device'` draws on the GPU (no
per-step H2D copy; same distribution, different stream).
"""
import torch

_DEVICE = [None]   # set_device(): where sampling tensors live; default = cuda if available
_NOISE = ['cpu']


def set_device(device):
    _DEVICE[0] = torch.device(device) if device is not None else None


def get_device():
    if _DEVICE[0] is not None:
        return _DEVICE[0]
    return torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')


def set_noise_source(kind):
    assert kind in ('cpu', 'device')
    _NOISE[0] = kind


def std_normal(size):
    """Standard normal tensor of `size` on the sampling device (reference: CPU draw, then .cuda())."""
    dev = get_device()
    if _NOISE[0] == 'device' and dev.type != 'cpu':
        return torch.randn(tuple(size), device=dev)
    return torch.normal(0, 1, size=tuple(size)).to(dev)


def calc_diffusion_hyperparams(T, beta_0, beta_T):
    """Linear beta schedule; Alpha_bar by sequential in-place f32 products, Sigma = sqrt(beta_tilde)."""
    Beta = torch.linspace(beta_0, beta_T, T)
    Alpha = 1 - Beta
    Alpha_bar = Alpha + 0
    Beta_tilde = Beta + 0
    for t in range(1, T):
        Alpha_bar[t] *= Alpha_bar[t - 1]
        Beta_tilde[t] *= (1 - Alpha_bar[t - 1]) / (1 - Alpha_bar[t])
    Sigma = torch.sqrt(Beta_tilde)
    return {"T": T, "Beta": Beta, "Alpha": Alpha, "Alpha_bar": Alpha_bar, "Sigma": Sigma}


def _expand_label(label, batch):
    if label is not None and isinstance(label, int):
        return (torch.ones(batch).long() * label).to(get_device())
    return label


def sampling(net, size, diffusion_hyperparams, print_every_n_steps=100, label=0, verbose=True, condition=None,
             return_multiple_t_slices=False, t_slices=[5, 10, 20, 50, 100, 200, 400, 600, 800],
             use_a_precomputed_XT=False, step=100, XT=None):
    """x_T ~ N(0,I); for t = T-1..0:  x <- (x - (1-a_t)/sqrt(1-abar_t) eps_theta(x,t)) / sqrt(a_t) (+ sigma_t z)."""
    dh = diffusion_hyperparams
    T, Alpha, Alpha_bar, Sigma = dh["T"], dh["Alpha"], dh["Alpha_bar"], dh["Sigma"]
    assert len(Alpha) == T and len(Alpha_bar) == T and len(Sigma) == T
    assert len(size) == 3
    dev = get_device()
    print('begin sampling, total number of reverse steps = %s' % T)
    slices = {}
    x = std_normal(size)
    label = _expand_label(label, size[0])
    if use_a_precomputed_XT:
        x = XT + Sigma[step] * std_normal(size)
        first = step - 1
    else:
        first = T - 1
    with torch.no_grad():
        for t in range(first, -1, -1):
            if verbose:
                print('t%d x max %.2f min %.2f' % (t, x.max(), x.min()))
            if t % print_every_n_steps == 0:
                print('reverse step: %d' % t, flush=True)
            ts = (t * torch.ones((size[0],))).to(dev)
            if condition is None:
                eps = net(x, ts=ts, label=label)
            else:
                eps = net(x, condition, ts=ts, label=label, use_retained_condition_feature=True)
            if verbose:
                print('t %d epsilon_theta max %.2f min %.2f' % (t, eps.max(), eps.min()))
            x = (x - (1 - Alpha[t]) / torch.sqrt(1 - Alpha_bar[t]) * eps) / torch.sqrt(Alpha[t])
            if return_multiple_t_slices and t in t_slices:
                slices[t] = x
            if t > 0:
                x = x + Sigma[t] * std_normal(size)
    if condition is not None:
        net.reset_cond_features()
    return (x, slices) if return_multiple_t_slices else x
