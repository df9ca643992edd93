"""Branch-decision comparison between the HIP path and the CPU oracle (checker-side helper: imported by tests/ and smoke() only).

The pass contains ~4.6 M (fixture size) to ~250 M (north-star size) ReLU / max-pool branch points.  Two exact-fp32
implementations with different summation orders differ by ~1e-7 in the pre-activations, so a pre-activation that lies
within rounding of 0 (or two pool candidates within rounding of each other) can be routed differently; ONE such flip
moves an individual small gradient tensor by 1e-4..2e-3 (e.g. 1/sqrt(808*512) for an FFN weight) although every kernel
is correct to 1e-6.  Parity of gradients is therefore asserted at 1e-4 when all branches agree, and the disagreeing
branches are required to be provable near-ties otherwise.
"""
import torch
import torch.nn.functional as F

NEAR_TIE = 2e-5


def oracle_trace(oracle, x, lens, y):
    """Run the oracle forward capturing conv pre-activations and FFN pre-activations."""
    pre = {}
    hooks = []
    for idx in (0, 2, 5, 7):
        hooks.append(oracle.conv[idx].register_forward_hook(lambda m, i, o, k=idx: pre.__setitem__('conv%d' % k, o.detach())))
    ffns = [('e%d' % i, l.pos_ffn) for i, l in enumerate(oracle.encoder.layers)] + \
           [('d%d' % i, l.pos_ffn) for i, l in enumerate(oracle.decoder.layers)]
    for tag, f in ffns:
        hooks.append(f.linear_1.register_forward_hook(lambda m, i, o, k=tag: pre.__setitem__(k + '.ff', o.detach())))
    out = oracle(x, lens, y)
    for h in hooks:
        h.remove()
    return out, pre


def _nhwc_to_ref(t):
    return t.permute(0, 3, 2, 1).cpu()


def disagreements(engine, pre):
    """-> (number of branch points decided differently, largest |margin| among them) for the LAST engine forward."""
    A = engine.arena
    n, worst = 0, 0.0

    def relu_site(mine_post, ref_pre):
        nonlocal n, worst
        bad = (mine_post > 0) != (ref_pre > 0)
        k = int(bad.sum())
        if k:
            n += k
            worst = max(worst, float(ref_pre[bad].abs().max()), float(mine_post[bad].abs().max()))

    def pool_site(p_mine, am_mine, ref_pre):
        nonlocal n, worst
        post = torch.relu(ref_pre)
        p_ref, idx = F.max_pool2d(post, 2, stride=2, return_indices=True)
        B, C, Fp, Tp = p_ref.shape
        T = ref_pre.shape[3]
        am = am_mine.long()
        f = torch.arange(Fp).view(1, 1, Fp, 1) * 2 + (am >> 1)
        t = torch.arange(Tp).view(1, 1, 1, Tp) * 2 + (am & 1)
        mine_idx = f * T + t
        sign_bad = (p_mine > 0) != (p_ref > 0)
        arg_bad = (mine_idx != idx) & (p_ref > 0) & (p_mine > 0)
        k = int(sign_bad.sum()) + int(arg_bad.sum())
        if k:
            n += k
            if int(sign_bad.sum()):
                worst = max(worst, float(p_ref[sign_bad].abs().max()), float(p_mine[sign_bad].abs().max()))
            if int(arg_bad.sum()):
                flat = post.flatten(2)
                mine_val = flat.gather(2, mine_idx.flatten(2)).view_as(p_ref)
                worst = max(worst, float((p_ref - mine_val)[arg_bad].abs().max()))

    relu_site(_nhwc_to_ref(A['y1']), pre['conv0'])
    pool_site(_nhwc_to_ref(A['p1']), _nhwc_to_ref(A['am1']), pre['conv2'])
    relu_site(_nhwc_to_ref(A['y5']), pre['conv5'])
    pool_site(_nhwc_to_ref(A['p2']), _nhwc_to_ref(A['am2']), pre['conv7'])
    for key, ref in pre.items():
        if key.endswith('.ff'):
            mine = A[key[:-3] + '.ff.h1'].cpu()
            relu_site(mine, ref.reshape(mine.shape))
    return n, worst


def gates_from_engine(engine):
    """The ReLU / max-pool decisions of the engine's LAST forward, in the layout oracle.refimpl's gate replay expects
    (conv maps as (B,C,F,T); FFN masks as (rows, inner)).  Everything is copied to the host."""
    A = engine.arena
    g = {'conv0': _nhwc_to_ref(A['y1'] > 0), 'conv5': _nhwc_to_ref(A['y5'] > 0),
         'am1': _nhwc_to_ref(A['am1']), 'pool1': _nhwc_to_ref(A['p1'] > 0),
         'am2': _nhwc_to_ref(A['am2']), 'pool2': _nhwc_to_ref(A['p2'] > 0)}
    for name, t in A.items():
        if name.endswith('.ff.h1'):
            g[name] = (t > 0).cpu()
    return g


def split_gates(g, nt):
    """gates of a task-batched forward (nt tasks' samples / rows stacked along dim 0) -> one dict per task"""
    parts = {k: torch.chunk(v, nt, dim=0) for k, v in g.items()}
    return [{k: parts[k][t] for k in g} for t in range(nt)]


class capture_gates:
    """with capture_gates(model) as log: ... -> log = [gates of every forward in the ORACLE's pass order (task 0 train, task 0
    validation, task 1 train, ...)] (every lane's engine).  Lanes enqueue in that order; a task-batched meta-iteration runs
    [training pass of all tasks, validation pass of all tasks], whose gates are split per task and interleaved on exit (one
    meta-iteration per capture).  The hook waits for the stream, so lanes are serialised while capturing (test only)."""

    def __init__(self, model):
        self.model, self.log, self._batched = model, [], []

    def __enter__(self):
        def hook(eng):
            torch.cuda.current_stream(eng.device).synchronize()
            g = gates_from_engine(eng)
            if eng.nt > 1:
                self._batched.append(split_gates(g, eng.nt))
            else:
                self.log.append(g)
        for e in self.model.engines:
            e.forward_hook = hook
        return self.log

    def __exit__(self, *exc):
        for e in self.model.engines:
            e.forward_hook = None
        if self._batched:
            nt = len(self._batched[0])
            for t in range(nt):
                for phase in self._batched:
                    self.log.append(phase[t])
        return False


def gates_from_oracle_trace(pre):
    """Same dict built from the ORACLE's own pre-activations (oracle_trace): replaying them must reproduce the free-running
    oracle exactly -- pins relu_replay / pool_replay and the 2*(f&1)+(t&1) window encoding against F.max_pool2d."""
    g = {'conv0': pre['conv0'] > 0, 'conv5': pre['conv5'] > 0}
    for key, name in (('conv2', '1'), ('conv7', '2')):
        z = pre[key]
        p, idx = F.max_pool2d(torch.relu(z), 2, stride=2, return_indices=True)
        T = z.shape[3]
        f, t = idx // T, idx % T
        g['am' + name] = (2 * (f & 1) + (t & 1)).to(torch.uint8)
        g['pool' + name] = p > 0
    for key, ref in pre.items():
        if key.endswith('.ff'):
            g[key + '.h1'] = ref.reshape(-1, ref.shape[-1]) > 0
    return g


def grad_tolerance(n_flips, worst_margin, clean=1e-4, flipped=1e-2):
    """1e-4 when HIP and oracle took identical branches; otherwise every disagreement must be a near-tie and the bound
    is the (documented) single-flip band."""
    if n_flips == 0:
        return clean
    assert worst_margin < NEAR_TIE, 'branch disagreement with margin %.3e is not a rounding near-tie' % worst_margin
    return flipped


class record_preactivations:
    """with record_preactivations(oracle) as log: ... -> log = [the pre-activation dict of oracle_trace for EVERY forward inside the
    block] (also with gates= replayed: the hooks sit on the convolution / linear modules, in front of the replayed decisions)."""

    def __init__(self, oracle):
        self.oracle, self.log, self.hooks = oracle, [], []

    def __enter__(self):
        o = self.oracle
        self.hooks.append(o.register_forward_pre_hook(lambda m, i: self.log.append({})))
        for idx in (0, 2, 5, 7):
            self.hooks.append(o.conv[idx].register_forward_hook(lambda m, i, out, k=idx: self.log[-1].__setitem__('conv%d' % k, out.detach())))
        ffns = [('e%d' % i, l.pos_ffn) for i, l in enumerate(o.encoder.layers)] + [('d%d' % i, l.pos_ffn) for i, l in enumerate(o.decoder.layers)]
        for tag, f in ffns:
            self.hooks.append(f.linear_1.register_forward_hook(lambda m, i, out, k=tag: self.log[-1].__setitem__(k + '.ff', out.detach())))
        return self.log

    def __exit__(self, *exc):
        for h in self.hooks:
            h.remove()
        return False


def replay_census(pre, gates):
    """(number of branch points where the decisions `gates` (the device's) differ from what the oracle's OWN pre-activations `pre`
    of the same forward would decide, largest margin among them).  With the device's decisions replayed the oracle walks the
    device's path, so every difference must be a rounding near-tie; a systematic gate / arg-max bug of the device path shows up
    here as many differences with large margins (the replayed oracle alone would mirror it)."""
    n, worst = 0, 0.0
    for key, gk in (('conv0', 'conv0'), ('conv5', 'conv5')):
        bad = (pre[key] > 0) != gates[gk]
        if int(bad.sum()):
            n += int(bad.sum())
            worst = max(worst, float(pre[key][bad].abs().max()))
    for key, nm in (('conv2', '1'), ('conv7', '2')):
        zpre = pre[key]
        post = torch.relu(zpre)
        p_ref, idx = F.max_pool2d(post, 2, stride=2, return_indices=True)
        T = zpre.shape[3]
        am = gates['am' + nm].long()
        Fp, Tp = p_ref.shape[2], p_ref.shape[3]
        f = torch.arange(Fp).view(1, 1, Fp, 1) * 2 + (am >> 1)
        t = torch.arange(Tp).view(1, 1, 1, Tp) * 2 + (am & 1)
        dev_idx = f * T + t
        dev_pos = gates['pool' + nm]
        sign_bad = dev_pos != (p_ref > 0)
        arg_bad = (dev_idx != idx) & dev_pos & (p_ref > 0)
        if int(sign_bad.sum()):
            n += int(sign_bad.sum())
            worst = max(worst, float(p_ref[sign_bad].abs().max()))
        if int(arg_bad.sum()):
            n += int(arg_bad.sum())
            dev_val = post.flatten(2).gather(2, dev_idx.flatten(2)).view_as(p_ref)
            worst = max(worst, float((p_ref - dev_val)[arg_bad].abs().max()))
    for key, ref in pre.items():
        if key.endswith('.ff'):
            z2 = ref.reshape(-1, ref.shape[-1])
            bad = (z2 > 0) != gates[key + '.h1']
            if int(bad.sum()):
                n += int(bad.sum())
                worst = max(worst, float(z2[bad].abs().max()))
    return n, worst
