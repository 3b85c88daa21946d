"""hipGraph-captured DDPM reverse loop (the MI355X execution model of `util.sampling`).

`util.sampling` (reference util.py:184-255) issues, per reverse step, ~60 native ops,
~150 1x1-conv GEMMs and several hundred small elementwise kernels from Python, draws
noise on the CPU and copies it (and the step index) to the GPU.  On MI355X a kernel
boundary costs ~1.5 us and an eager launch ~3.5 us of host time, so the loop is
launch-bound long before it is compute-bound.

Here ONE cached-condition reverse step
        eps = net(x, cond, ts, label);  x <- (x - c1[t] eps) * c2[t] + sigma[t] z
is captured into a hipGraph once per batch shape and replayed T-1 times:
  * the step index lives on the device and is decremented inside the graph; the
    coefficients (1-alpha_t)/sqrt(1-abar_t), 1/sqrt(alpha_t), sigma_t (sigma_0 := 0, the
    reference adds no noise at t=0) are gathered from device tables;
  * noise is either drawn on the device inside the graph (Philox, `noise='device'`) or,
    for seed-parity with the reference, drawn on the CPU default generator and copied
    into a static buffer before each replay (`noise='cpu'`);
  * the first step of a batch (condition branch + global PointNet, results retained by
    the network) runs eagerly, as does graph capture itself.
The arithmetic per step is the reference's, in the same order.

Two captured forms of the step (round 5).  The fused network evaluates a neighbourhood that ball_query filled with K
copies of one point ONCE (fused_network.DEDUP) -- the rule while x_t is noise, i.e. for most of a
reverse process -- at the price of a per-query chain beside every per-neighbour launch; on a finished surface, where
balls are full, that chain is pure overhead.  The sampler therefore captures the step BOTH ways (`once` and `whole`,
each on first need) and picks per step on the host: every step counts, on the device, the tiles a deduplicated step
walks / would walk (pdr_dedup_prepare / pdr_dedup_probe), the step's last kernel publishes the two counters into
pinned host memory, and before launching step i the host waits for step i-2 (an event: the queue never runs dry, the
GPU never waits) and reads the slot THAT step published (a four-slot ring tagged with the step counter) -- `once` while
the walked share stays below WHOLE_ABOVE, `whole` above it.  The choice is a function of the data, not of timing: a run
repeated from the same seed replays the same forms.  Same results either way up to fp32 summation order of GroupNorm
moments (1e-6 per step; over many steps a near-tie of an FPS pick can resolve differently between the forms, so
`neighbourhoods='once'` / `'whole'` are the settings under which two DIFFERENT switch thresholds would still agree).
"""
import torch

# The first, uncached step of a batch (condition branch + the step itself: ~600 launches, 16-19 ms when the host submits
# them one by one) as a hipGraph too: captured at the first batch of a shape, BEFORE the cached step's graphs, so that
# the retained condition features the cached steps read ARE the tensors this graph writes (no copy between batches).
# Fused network + native update only; False: the eager first step of rounds 1-5 (a variant of the tests).
FIRST_STEP_GRAPH = True


class GraphedReverseSampler:
    # neighbourhoods='adaptive': walked share (tiles walked / tiles of the deduplicable blocks) above which the step
    # with every neighbourhood evaluated is replayed.  bench.py's `trajectory` leg times both forms over the share.
    # (MI355X, B = 32, x_t = q_sample(torus, t): the two forms cross at a walked share of ~0.6 -- 0.40: 8.11 vs 8.90 ms,
    # 0.53: 8.57 vs 8.87, 0.68: 9.12 vs 8.86, 0.96: 9.92 vs 8.87; profiles/r5_bench.json `trajectory`)
    WHOLE_ABOVE = 0.6

    def __init__(self, net, diffusion_hyperparams, noise='device', use_graph=True, neighbourhoods='adaptive'):
        """neighbourhoods: 'adaptive' (default; see the module docstring), 'once' / 'whole' = one form for every step
        (networks without the switch -- the layer-by-layer module -- have one form anyway)."""
        assert noise in ('cpu', 'device')
        assert neighbourhoods in ('adaptive', 'once', 'whole')
        self.net = net
        self.noise = noise
        self.use_graph = use_graph
        self.neighbourhoods = neighbourhoods if hasattr(net, "dedup") else 'once'
        self._graphs = {}
        self._mode = 'once'
        self.mode_counts = {'once': 0, 'whole': 0}
        dh = diffusion_hyperparams
        self.T = int(dh["T"])
        self.device = next(net.parameters()).device
        A, Ab, S = dh["Alpha"].float().cpu(), dh["Alpha_bar"].float().cpu(), dh["Sigma"].float().cpu()
        # evaluated exactly like `(1-Alpha[t])/torch.sqrt(1-Alpha_bar[t])` and `torch.sqrt(Alpha[t])`
        self.c_eps = ((1 - A) / torch.sqrt(1 - Ab)).to(self.device)
        self.sqrt_alpha = torch.sqrt(A).to(self.device)
        sig = S.clone()
        sig[0] = 0.0
        self.sigma = sig.to(self.device)
        self._sigma_raw = S.to(self.device)            # Sigma[step] of the restart option (sigma_0 not zeroed)
        self._key = None
        self._step_table = None                        # built on first use (needs the subclass's time table)
        self._first = None                             # (mode, graph, retained tensors) of the captured first step

    @property
    def _graph(self):
        """(tests / tools) the captured step in use."""
        return self._graphs.get(self._mode)

    def _network_times(self):
        """(T,) network time inputs indexed by the device step counter: float(t) (DDPM); subclasses: their tau table."""
        return torch.arange(self.T, dtype=torch.float32, device=self.device)

    # ------------------------------------------------------------------ one step
    NOISE_AT_LAST_STEP = False     # util.sampling draws no noise at t = 0

    def _timestep(self, t):
        """Network time input of the step whose device counter is `t` ((1,) int64)."""
        return t.to(torch.float32)

    UPDATE_MODE = 0                # pdr_reverse_update: 0 = DDPM ancestral step, 1 = FastDPM / DDIM-style step

    def _tables(self):
        return self.c_eps, self.sqrt_alpha, self.sigma

    def _update(self, x, eps, t, z):
        """Reference expression (util.py:246-250) in PyTorch ops -- the host-logic / CPU form; on the GPU the same
        operations run as ONE native launch (`_update_native`, bit-identical)."""
        # index_select keeps the lookup on the device (tensor[t] with a 0-d index would call .item())
        c_eps, sqrt_a, sigma = (tab.index_select(0, t) for tab in self._tables())
        return (x - c_eps * eps) / sqrt_a + sigma * z

    def _native(self):
        x = self._x
        return x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[2] == 3

    def _step_native(self, eps):
        """One pdr_reverse_step launch: the update (bit-identical to `_update`), the noise of `noise='device'` drawn
        inside the kernel, `t -= 1` and the next step's network time input -- all on the device, no other kernel.
        eps may be the strided (.., 4)-row view the fused network's last layer writes."""
        from .. import _lib
        x = self._x
        B, N, _ = x.shape
        assert eps.dtype == torch.float32 and eps.device == x.device and tuple(eps.shape) == tuple(x.shape), \
            "eps must be a float32 tensor of x's shape on x's device"
        if not (eps.stride(2) == 1 and eps.stride(0) == N * eps.stride(1)):
            eps = eps.contiguous()
        a, b, c = self._tables()
        device_noise = self.noise == 'device'
        z = None if device_noise else self._z
        assert z is None or (z.is_contiguous() and z.device == x.device and z.dtype == torch.float32)
        tab = self._ts_table()
        with torch.cuda.device(x.device):
            probe = getattr(self.net, "probe", None)
            _lib.check(_lib.load().pdr_reverse_step(
                x.data_ptr(), eps.data_ptr(), eps.stride(1), None if z is None else z.data_ptr(), a.data_ptr(),
                b.data_ptr(), c.data_ptr(), self._t.data_ptr(), None if tab is None else tab.data_ptr(),
                self._ts.data_ptr(), self._rng.data_ptr() if device_noise else None, self._ticket.data_ptr(), B * N,
                self.UPDATE_MODE, None if probe is None else probe.data_ptr(),
                None if probe is None else self._probe_host.data_ptr(),
                torch.cuda.current_stream(x.device).cuda_stream), "reverse_step")

    def _ts_table(self):
        """Device table of network time inputs indexed by the step counter, or None = float(t) (DDPM)."""
        return None

    def _step(self, keep_slice=None):
        """One reverse step.  keep_slice: a list -- the step runs in PyTorch ops and appends the cloud BEFORE the noise
        is added, what util.sampling stores for `return_multiple_t_slices` (util.py:246-248)."""
        t = self._t                                   # (1,) int64 on device, counts down to 0
        native = self._native() and keep_slice is None
        B = self._x.shape[0]
        # native: the time input of this step was published by the previous step's pdr_reverse_step (or by begin())
        ts = (self._ts if native else self._timestep(t)).expand(B)
        strided = native and hasattr(self.net, "return_strided_eps")
        saved_flag = getattr(self.net, "return_strided_eps", None)
        if strided:
            self.net.return_strided_eps = True         # fused network: hand over its 4-float output rows as a view
        fused = hasattr(self.net, "dedup")
        if fused:
            saved_net = self.net.dedup, self.net.probe, self.net.step_table
            self.net.dedup = self._mode == 'once'
            # the probe counters travel only with the native step (its last kernel publishes and resets them)
            self.net.probe = self._probe if (native and self.neighbourhoods == 'adaptive') else None
            # (None when the network cannot build the table -- NATIVE_EMBED off, a t_dim / bank width outside the
            # kernels' contract: the network then runs its per-step chain; ADVICE r5: a (None, t) pair crashed it)
            tab = self._table() if native else None
            self.net.step_table = (tab, self._t) if tab is not None else None
        try:
            eps = self.net(self._x, self._cond, ts=ts, label=self._label, use_retained_condition_feature=True)
            if native:
                self._step_native(eps)
                return
        finally:
            if strided:
                self.net.return_strided_eps = saved_flag
            if fused:
                self.net.dedup, self.net.probe, self.net.step_table = saved_net
        z = torch.randn_like(self._x) if self.noise == 'device' else self._z
        if keep_slice is not None:
            if self.UPDATE_MODE != 0:
                raise NotImplementedError("t-slices exist for the DDPM loop only (util.sampling)")
            c_eps, sqrt_a, sigma = (tab.index_select(0, t) for tab in self._tables())
            x = (self._x - c_eps * eps) / sqrt_a
            keep_slice.append(x.clone())
            self._x.copy_(x + sigma * z)
        else:
            self._x.copy_(self._update(self._x, eps, t, z))
        self._t.sub_(1)
        if self._x.is_cuda:
            self._ts.copy_(self._timestep(self._t.clamp(min=0)))   # what pdr_reverse_step publishes on the native path

    def _prepare(self, size, condition, label):
        key = (tuple(size), tuple(condition.shape), None if label is None else tuple(label.shape))
        if self._key != key:
            self._graphs = {}
            self._first = None
            self._key = key
            self._x = torch.empty(size, device=self.device)
            self._z = torch.empty(size, device=self.device)
            self._cond = torch.empty_like(condition, device=self.device)
            self._label = None if label is None else torch.empty_like(label, device=self.device)
            self._t = torch.zeros((1,), dtype=torch.int64, device=self.device)
            self._ts = torch.zeros((1,), dtype=torch.float32, device=self.device)     # network time input of the step
            self._rng = torch.zeros((2,), dtype=torch.int64, device=self.device)      # Philox key, draw number
            self._ticket = torch.zeros((1,), dtype=torch.int32, device=self.device)
            # neighbourhood probe: device counters (the network's geometry launches add to them, the step's last kernel
            # publishes and zeroes them) and their pinned, device-visible host copy
            self._probe = torch.zeros((2,), dtype=torch.int32, device=self.device)
            # (a ring of four slots {walked, tiles, step counter, written}: slot t & 3, pdr_reverse_step)
            self._probe_host = torch.zeros((16,), dtype=torch.int32).pin_memory() if self.device.type == "cuda" \
                else torch.zeros((16,), dtype=torch.int32)
            self._events = []
        self._cond.copy_(condition)
        if label is not None:
            self._label.copy_(label)

    def _table(self):
        """The network's step-embedding table for this sampler's schedule (built once; None: per-step chain)."""
        if self._step_table is None:
            build = getattr(self.net, "build_step_table", None)
            tab = build(self._network_times()) if build is not None else None
            self._step_table = (tab,)
        return self._step_table[0]

    def _capture(self):
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        state = (self._x, self._t, self._ts, self._rng, self._probe)
        saved = [v.clone() for v in state]
        self._table()                                  # (built outside the capture, if it was not yet)

        def restore():
            for v, w in zip(state, saved):
                v.copy_(w)
        with torch.cuda.stream(s):
            self._step()                               # warm-up on the side stream (allocator, lazy init)
        torch.cuda.current_stream(self.device).wait_stream(s)
        restore()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._step()
        restore()
        first = not self._graphs
        self._graphs[self._mode] = g
        # the graph has baked in the addresses of the retained condition features (the second form is captured on the
        # same tensors: by then the network already points at them)
        if first:
            self._static_cache = self._cache_tensors()

    _CACHE_ATTRS = ("l_uvw", "encoder_cond_features", "decoder_cond_features")

    def _cache_tensors(self):
        net = getattr(self.net, "net", self.net)       # FusedCloudConditionNet wraps the torch module
        items = [net.global_feature]
        for name in self._CACHE_ATTRS:
            items.extend(getattr(net, name) or [])
        return items

    def _adopt_cache(self):
        """A new batch produced fresh retained features; move their VALUES into the tensors the captured
        graph reads and point the network back at those."""
        net = getattr(self.net, "net", self.net)
        fresh = self._cache_tensors()
        assert len(fresh) == len(self._static_cache)
        for dst, src in zip(self._static_cache, fresh):
            if dst is not None:
                dst.copy_(src)
        it = iter(self._static_cache)
        net.global_feature = next(it)
        for name in self._CACHE_ATTRS:
            cur = getattr(net, name)
            if cur is not None:
                setattr(net, name, [next(it) for _ in cur])

    # ------------------------------------------------------------------ public
    @torch.no_grad()
    def begin(self, size, condition, label=None, x_T=None, start_step=None, keep_slice=None):
        """Load a batch: x_T (drawn like the reference if not given), condition, labels; run the first
        (uncached) reverse step eagerly so the network retains its condition features."""
        self._prepare(size, condition, label)
        self.net.reset_cond_features()
        if x_T is None:
            x_T = torch.randn(size, device=self.device) if self.noise == 'device' else torch.normal(0, 1, size=size)
        self._x.copy_(x_T)
        t0 = self.T - 1 if start_step is None else int(start_step)
        self._t.fill_(t0)
        self._ts.copy_(self._timestep(self._t))
        if self.noise == 'device':
            # key of this batch's in-kernel normal stream: drawn from the CPU default generator (torch.manual_seed
            # makes a run reproducible); the draw number restarts at 0
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            self._rng.copy_(torch.tensor([seed, 0], dtype=torch.int64))
        self.remaining = t0 + 1
        self._mode = 'whole' if self.neighbourhoods == 'whole' else 'once'
        self._probe.zero_()
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()   # (nothing of the previous batch still publishes)
        self._probe_host.zero_()                       # no slot of the previous batch can be mistaken for this one's
        self._events = []
        first_graph = self.use_graph and FIRST_STEP_GRAPH and keep_slice is None and self._native() and \
            hasattr(self.net, "sync_condition") and self.device.type == "cuda"
        replayed = False
        if first_graph and self._first is not None and self._first[0] == self._mode:
            self._replay_first()                       # first step: condition branch + step, one graph replay
            replayed = True
        else:
            state = (self._x, self._t, self._ts, self._rng, self._probe)
            saved = [v.clone() for v in state] if first_graph and not self._graphs and self._first is None else None
            self._advance_eager(keep_slice)            # first step: condition branch runs and is retained
            if saved is not None:
                self._capture_first(state, saved)      # (once per shape: ahead of the cached step's graphs)
        if self.neighbourhoods == 'adaptive' and self.device.type == "cuda":
            # once per batch: the first step's probe decides the form of the second (a restart from a stored x^step
            # begins on a surface); later steps read the probe of the step two before them, by its step counter (a
            # first step that ran in PyTorch ops -- keep_slice -- published nothing: the default form stays)
            torch.cuda.current_stream(self.device).synchronize()
            self._pick_mode(t0)
        if self._graphs and self._first is None:
            self._adopt_cache()
        if hasattr(self.net, "sync_condition") and not replayed:
            self.net.sync_condition()                  # fused network: refresh its channel-last copies in place

    def _capture_first(self, state, saved):
        """The first step of a batch as a graph.  Called right after the eager first step of the FIRST batch of a shape
        (the warm-up) with the state that step started from: capture (fresh condition features are allocated from the
        graph's pool -- they stay where they are from now on), restore the state, replay once so that the retained
        tensors hold this batch's values."""
        def restore():
            for v, w in zip(state, saved):
                v.copy_(w)
        restore()
        self.net.reset_cond_features()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._step()
        restore()
        g.replay()
        net = getattr(self.net, "net", self.net)
        self._first_lens = {name: (None if getattr(net, name) is None else len(getattr(net, name)))
                            for name in self._CACHE_ATTRS}
        self._first = (self._mode, g, self._cache_tensors())

    def _replay_first(self):
        mode, g, tensors = self._first
        self._draw_cpu_noise()
        self.net.reset_cond_features()
        g.replay()
        # the Python side of an eager first step: the network points at its retained features, the fused wrapper knows
        # its channel-last copies and class-embedding rows are those of this batch
        net = getattr(self.net, "net", self.net)
        it = iter(tensors)
        net.global_feature = next(it)
        lens = self._first_lens
        for name in self._CACHE_ATTRS:
            n = lens[name]
            setattr(net, name, None if n is None else [next(it) for _ in range(n)])
        self.net._synced = True
        if self._label is not None:
            self.net._label_key = (self._label, self._label._version)
        self.remaining -= 1

    def _draw_cpu_noise(self):
        # reference order: one draw per step with t > 0, none at t = 0 (FastDPM: every step)
        if self.noise == 'cpu' and (self.remaining > 1 or self.NOISE_AT_LAST_STEP):
            self._z.copy_(torch.normal(0, 1, size=tuple(self._x.shape)))
        elif self.noise == 'cpu':
            self._z.zero_()

    def _advance_eager(self, keep_slice=None):
        self._draw_cpu_noise()
        self._step(keep_slice)
        self.remaining -= 1

    def _pick_mode(self, t):
        """Form of the next step from the probe the step with counter `t` published (pinned host memory, slot t & 3 of
        the ring, tagged with t: the caller has waited for that step, and a later step that has already overwritten the
        slot -- it cannot, with two steps in flight and four slots -- or a step that published nothing leaves the form
        as it is).  The choice therefore depends on the data only, not on how far the device has run ahead: two runs
        from one seed replay the same forms."""
        slot = self._probe_host[4 * (t & 3):4 * (t & 3) + 4].tolist()
        walked, total, tag, written = slot
        if written and tag == t and total > 0:
            self._mode = 'whole' if walked > self.WHOLE_ABOVE * total else 'once'
            self.walked_share = walked / total
        elif not (written and tag == t):
            self.walked_share = None

    @torch.no_grad()
    def advance(self, n=1):
        """Run n more reverse steps (graph replay)."""
        adaptive = self.neighbourhoods == 'adaptive'
        for _ in range(n):
            assert self.remaining > 0, "reverse process already finished"
            if adaptive and self.device.type == "cuda":
                # wait for the step before the previous one (keeps two steps queued), then read what it published
                if len(self._events) >= 2:
                    ev, t_ev = self._events.pop(0)
                    ev.synchronize()
                    self._pick_mode(t_ev)
            if not self.use_graph:
                self._advance_eager()
            else:
                if self._mode not in self._graphs:
                    self._capture()
                self._draw_cpu_noise()
                self._graphs[self._mode].replay()
                self.remaining -= 1
            self.mode_counts[self._mode] += 1
            if adaptive and self.device.type == "cuda":
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                self._events.append((ev, self.remaining))       # (the step just queued ran with counter `remaining`)

    @torch.no_grad()
    def finish(self):
        self.advance(self.remaining)
        out = self._x.clone()
        self.net.reset_cond_features()
        return out

    @torch.no_grad()
    def sample(self, size, condition, label=None, x_T=None, return_multiple_t_slices=False,
               t_slices=(5, 10, 20, 50, 100, 200, 400, 600, 800), use_a_precomputed_XT=False, step=100, XT=None):
        """Equivalent of util.sampling(net, size, dh, label=label, condition=condition, ...) incl. its two options
        (util.py:217-222, 246-248):
          use_a_precomputed_XT: restart from a stored x^step: x = XT + Sigma[step] z, then steps step-1 ... 0;
          return_multiple_t_slices: also return {t: cloud after the update of step t, BEFORE its noise} for t in
          t_slices -- those few steps run through PyTorch ops (eagerly), all others as graph replays."""
        start = None
        if use_a_precomputed_XT:
            if XT is None or not 1 <= int(step) <= self.T - 1:
                raise ValueError("use_a_precomputed_XT needs XT (the stored x^step) and 1 <= step <= T-1")
            if self.noise == 'cpu':
                torch.normal(0, 1, size=size)          # util.sampling draws (and discards) x_T first: same CPU stream
            z = torch.randn(size, device=self.device) if self.noise == 'device' else torch.normal(0, 1, size=size).to(self.device)
            x_T = XT.to(self.device) + self._sigma_raw[step] * z
            start = step - 1
        slices = {}
        want = set(int(t) for t in t_slices) if return_multiple_t_slices else set()
        t0 = (self.T - 1) if start is None else start
        keep = [] if t0 in want else None
        self.begin(size, condition, label, x_T, start_step=start, keep_slice=keep)
        if keep:
            slices[t0] = keep[0]
        while self.remaining > 0:
            t = self.remaining - 1
            if t in want:
                keep = []
                self._advance_eager(keep)
                slices[t] = keep[0]
            else:
                self.advance(1)
        out = self.finish()
        return (out, slices) if return_multiple_t_slices else out


class GraphedFastSampler(GraphedReverseSampler):
    """hipGraph-captured FastDPM loop (util_fastdpmv2.fast_sampling_function_v2 :455-476: VAR_sampling :307-381 /
    STEP_sampling :384-452): S << T network calls at (fractional) times tau_i with the DDIM-style update
        x <- x sqrt(a'/a) + ( (sqrt(1 - a' - s^2) - sqrt(1 - a) sqrt(a'/a)) eps + s z ),   a' := 1, s := 0 at the end.
    The per-step constants are evaluated on the host exactly like the reference (float32 torch scalars), stored in
    device tables indexed by the same down-counting device step counter as the DDPM loop."""

    NOISE_AT_LAST_STEP = True      # _ddim_update draws std_normal on every step, also when sigma = 0

    def __init__(self, net, diffusion_hyperparams, diffusion_config, length=50, sampling_method='var',
                 schedule='quadratic', kappa=0.0, noise='device', use_graph=True, neighbourhoods='adaptive'):
        from . import util_fastdpmv2 as F
        super().__init__(net, diffusion_hyperparams, noise=noise, use_graph=use_graph, neighbourhoods=neighbourhoods)
        dh = diffusion_hyperparams
        assert sampling_method in ('var', 'step') and schedule in ('quadratic', 'linear') and 0.0 <= kappa <= 1.0
        if sampling_method == 'var':
            eta = F.get_VAR_noise(length, diffusion_config, schedule)
            steps = F._precompute_VAR_steps(dh, eta)
            gamma_bar = F._gamma_bar(eta)
            S = len(gamma_bar)
            alpha_of = lambda i: gamma_bar[S - 1 - i]
            assert abs(steps[-1]) < 0.1
        else:
            steps = sorted(list(F.get_STEP_step(length, diffusion_config, schedule)), reverse=True)
            abar = dh["Alpha_bar"].cpu()
            alpha_of = lambda i: abar[steps[i]]
            assert steps[-1] == 0
        n = len(steps)
        scale, c, sig, tau = [], [], [], []
        for i in range(n):
            a_cur = alpha_of(i)
            if i == n - 1:
                a_next, sigma = torch.tensor(1.0), torch.tensor(0.0)
            else:
                a_next = alpha_of(i + 1)
                sigma = kappa * torch.sqrt((1 - a_next) / (1 - a_cur) * (1 - a_cur / a_next))
            scale.append(torch.sqrt(a_next / a_cur))
            c.append(torch.sqrt(1 - a_next - sigma ** 2) - torch.sqrt(1 - a_cur) * torch.sqrt(a_next / a_cur))
            sig.append(torch.as_tensor(sigma, dtype=torch.float32))
            tau.append((steps[i] * torch.ones((1,)))[0])           # float32, as `tau * torch.ones((B,))`
        # the device counter runs S-1 ... 0: entry t of a table belongs to step i = S-1-t
        rev = lambda xs: torch.stack([torch.as_tensor(v, dtype=torch.float32) for v in xs[::-1]]).to(self.device)
        self.f_scale, self.f_c, self.f_sigma, self.f_tau = rev(scale), rev(c), rev(sig), rev(tau)
        self.T = n

    UPDATE_MODE = 1

    def _timestep(self, t):
        return self.f_tau.index_select(0, t)

    def _ts_table(self):
        return self.f_tau

    def _network_times(self):
        return self.f_tau

    def _tables(self):
        return self.f_scale, self.f_c, self.f_sigma

    def _update(self, x, eps, t, z):
        scale, c, sigma = (tab.index_select(0, t) for tab in self._tables())
        x = x * scale
        return x + (c * eps + sigma * z)
