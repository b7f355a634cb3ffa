"""The hand-written bf16 MFMA MLP kernels (csrc/pnr_mlp.h, pioneer_amd/mlp.py) against torch.

Two references on the same weights and inputs: (a) plain float32 torch autograd of the ActorCritic module — the
semantics; tolerance = what bf16 operands cost (relative L2 <= 1.5e-2 forward, 3e-2 gradients); (b) a float64
restatement that rounds to bf16 exactly where the kernels do (weights, inputs, stored activations, propagated
gradients) — the kernels' arithmetic; relative L2 <= 2e-3 (float32 accumulation order, the exp2-based tanh).
Batches 1, 4 099 (ragged tiles and slices) and 131 072 (the large minibatch), with and without the row gather and
the observation filter."""
import pytest
import torch

pytestmark = pytest.mark.gpu
NAMES = ["w1", "b1", "w2", "b2", "w3", "b3"] * 2


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def bf(x):
    return x.to(torch.bfloat16).double()


def make(B, seed, rows=None, with_filter=False):
    from pioneer_amd.ppo import ActorCritic, PPOConfig
    from pioneer_amd.mlp import HipMLP
    dev = torch.device("cuda", 0)
    torch.manual_seed(seed)
    model = ActorCritic(PPOConfig()).to(dev)
    with torch.no_grad():                                   # heads of visible size, non-zero biases
        for net in (model.policy, model.value):
            for l in net:
                if isinstance(l, torch.nn.Linear):
                    l.bias.normal_(0, 0.1)
        model.policy[4].weight.mul_(30.0)
    g = torch.Generator(device=dev).manual_seed(seed + 1)
    rows = rows or B
    obs = torch.randn(rows, 137, generator=g, device=dev) * 2.0 + 0.3
    idx = torch.randperm(rows, generator=g, device=dev)[:B].contiguous() if rows != B else None
    filt = None
    if with_filter:
        loc = torch.randn(137, generator=g, device=dev) * 0.5
        inv = torch.rand(137, generator=g, device=dev) + 0.5
        hi = torch.full((137,), 2.5, device=dev)
        filt = (loc, inv, -hi, hi)
    return model, HipMLP(model, B, dev), obs, idx, filt


def net_input(obs, idx, filt):
    x = obs if idx is None else obs[idx]
    if filt is not None:
        x = torch.clamp((x - filt[0]) * filt[1], min=filt[2], max=filt[3])
    return x


def head_grads(B, dev, seed=5):
    g = torch.Generator(device=dev).manual_seed(seed)
    g_head = torch.randn(2, B, 16, generator=g, device=dev) * (1.0 / B)
    g_head[0, :, 12:] = 0
    g_head[1, :, 1:] = 0                                    # the loss kernel never puts gradient on the padding rows
    return g_head


def emulate(model, x, g_head):
    """float64 restatement with the kernels' bf16 rounding points.  Returns heads [2][B][16] and 12 gradients."""
    heads, grads = [], []
    xb = bf(x)
    for n, net in enumerate((model.policy, model.value)):
        W1, b1, W2, b2, W3, b3 = [t.detach() for l in net if isinstance(l, torch.nn.Linear) for t in (l.weight, l.bias)]
        h1 = bf(torch.tanh(xb @ bf(W1).t() + b1.double()))
        h2 = bf(torch.tanh(h1 @ bf(W2).t() + b2.double()))
        n3 = W3.shape[0]
        head = torch.zeros(x.shape[0], 16, dtype=torch.float64, device=x.device)
        head[:, :n3] = h2 @ bf(W3).t() + b3.double()
        heads.append(head)
        gb = bf(g_head[n])[:, :n3]
        dz2 = bf((gb @ bf(W3)) * (1 - h2 * h2))
        dz1 = bf((dz2 @ bf(W2)) * (1 - h1 * h1))
        grads += [dz1.t() @ xb, dz1.sum(0), dz2.t() @ h1, dz2.sum(0), gb.t() @ h2, gb.sum(0)]
    return torch.stack(heads), grads


@pytest.mark.parametrize("B,rows,with_filter", [(1, None, False), (4099, 6000, True), (131072, None, False), (32768, 40000, True)])
def test_forward_and_backward_match_torch(B, rows, with_filter):
    model, mlp, obs, idx, filt = make(B, seed=B % 97, rows=rows, with_filter=with_filter)
    x = net_input(obs, idx, filt)
    g_head = head_grads(B, obs.device)

    # --- kernels
    hp, hv = mlp.apply(obs, idx, filt)
    assert hp.shape == (B, 16) and hv.shape == (B, 16)
    got = torch.autograd.grad((hp * g_head[0]).sum() + (hv * g_head[1]).sum(), mlp.params)
    assert bool((hp[:, 12:] == 0).all()) and bool((hv[:, 1:] == 0).all())   # zero rows of the padded heads

    # --- (a) float32 torch autograd
    ref_p, ref_v = model.policy(x), model.value(x)
    ref = torch.autograd.grad((ref_p * g_head[0, :, :12]).sum() + (ref_v * g_head[1, :, :1]).sum(), mlp.params)
    assert rel(hp[:, :12], ref_p) < 1.5e-2 and rel(hv[:, :1], ref_v) < 1.5e-2
    for a, b, name in zip(got, ref, NAMES):
        assert rel(a, b) < 3e-2, (name, rel(a, b))

    # --- (b) the kernels' arithmetic restated in float64 with bf16 rounding points
    e_heads, e_grads = emulate(model, x, g_head)
    assert rel(hp, e_heads[0]) < 2e-3 and rel(hv, e_heads[1]) < 2e-3
    for a, b, name in zip(got, e_grads, NAMES):
        assert rel(a, b) < 2e-3, (name, rel(a, b))


def test_nograd_forward_equals_training_forward_and_respects_bounds():
    """The sampling path (no saved activations) gives the same heads and writes nothing behind its output."""
    B = 3000
    model, mlp, obs, idx, filt = make(B, seed=3, with_filter=True)
    hp, hv = mlp.apply(obs, idx, filt)
    guard = torch.full((2 * B * 16 + 4096,), 7.0, device=obs.device)
    hb = guard[:2 * B * 16].view(2, B, 16)
    h = mlp.forward_nograd(obs, idx, filt, out=hb)
    assert torch.equal(h[0], hp) and torch.equal(h[1], hv)
    assert bool((guard[2 * B * 16:] == 7.0).all())


@pytest.mark.parametrize("B,clip", [(1, True), (4099, True), (16384, False)])
def test_act_equals_the_torch_glue_on_the_same_heads(B, clip):
    """pnr_mlp_act = forward_nograd + the sampler's torch glue (mean / clamp(log_std) / value slices, mean + exp(log_std)
    * noise, clip to the action space): heads bit for bit, the draw to float32 rounding of exp (torch.exp vs expf)."""
    model, mlp, obs, _, filt = make(B, 21, with_filter=True)
    dev = obs.device
    mlp.pack()
    heads = mlp.forward_nograd(obs, None, filt)
    g = torch.Generator(device=dev).manual_seed(3)
    noise = torch.randn(B, 6, generator=g, device=dev)
    a_max = torch.rand(6, generator=g, device=dev) * 2.0 + 0.5
    f32 = dict(dtype=torch.float32, device=dev)
    out = {k: torch.full((B, 6), float("nan"), **f32) for k in ("mean", "log_std", "actions", "env_actions")}
    values = torch.full((B,), float("nan"), **f32)
    head2 = torch.full((2, B, 16), float("nan"), **f32)
    xs = torch.full((B, 144), float("nan"), dtype=torch.bfloat16, device=dev)
    mlp.act(obs, filt, noise, a_max if clip else None, mean=out["mean"], log_std=out["log_std"], values=values,
            actions=out["actions"], env_actions=out["env_actions"] if clip else None, head=head2, xs_out=xs)
    torch.cuda.synchronize()
    assert torch.equal(head2, heads)
    # the saved net input: the filtered observation rounded to bf16, seven zero columns of padding
    assert torch.equal(xs[:, :137], net_input(obs, None, filt).to(torch.bfloat16)) and float(xs[:, 137:].float().abs().max()) == 0.0
    mean, log_std = heads[0, :, :6], torch.clamp(heads[0, :, 6:12], -20.0, 2.0)
    assert torch.equal(out["mean"], mean) and torch.equal(out["log_std"], log_std) and torch.equal(values, heads[1, :, 0])
    act = torch.addcmul(mean, torch.exp(log_std), noise)
    # |d act| <= 2 ulp of exp(log_std) * |noise| + 1 ulp of the sum
    bound = 2.4e-7 * (torch.exp(log_std) * noise.abs()) + 1.2e-7 * act.abs() + 1e-30
    assert bool(((out["actions"] - act).abs() <= bound).all()), float((out["actions"] - act).abs().max())
    if clip:
        assert torch.equal(out["env_actions"], torch.minimum(torch.maximum(out["actions"], -a_max), a_max))
        assert bool((out["env_actions"].abs() <= a_max).all()) and bool((out["actions"].abs() > a_max).any())
    else:
        assert bool(torch.isnan(out["env_actions"]).all())           # not written without a_max


def test_graph_replay_of_the_learner_kernels_is_exact():
    """Captured once, replayed on fresh inputs: bit-identical heads and gradients to eager launches, replay after
    replay (no semaphores, no atomics, no library workspaces in these kernels; fixed-order slab reduction)."""
    B = 8192
    model, mlp, obs, idx, filt = make(B, seed=11)
    dev = obs.device
    g_head = head_grads(B, dev)
    static_obs = obs.clone()

    def run():
        hp, hv = mlp.apply(static_obs)
        return hp, hv, torch.autograd.grad((hp * g_head[0]).sum() + (hv * g_head[1]).sum(), mlp.params)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        hp, hv, grads = run()
    gen = torch.Generator(device=dev).manual_seed(0)
    for it in range(4):
        static_obs.copy_(torch.randn(B, 137, generator=gen, device=dev))
        g_head.copy_(head_grads(B, dev, seed=100 + it))
        gr.replay()
        ehp, ehv, egrads = run()
        assert torch.equal(hp, ehp) and torch.equal(hv, ehv)
        for a, b in zip(grads, egrads):
            assert torch.equal(a, b)


def _record(B, dev, seed=9):
    from pioneer_amd.ppo import gaussian_logp
    g = torch.Generator(device=dev).manual_seed(seed)
    R = lambda *s: torch.randn(*s, generator=g, device=dev)   # noqa: E731
    act, mean, ls = R(B, 6), 0.1 * R(B, 6), 0.1 * R(B, 6)
    return {"actions": act, "mean": mean, "log_std": ls, "logp": gaussian_logp(act, mean, ls), "values": R(B), "adv": R(B), "vtarg": R(B)}


def _unpack_flat(flat):
    """The padded gradient bucket [2][107 024] as the twelve parameter-shaped gradients (policy six, value six)."""
    E = flat.numel() // 2
    out = []
    for n in range(2):
        f = flat[n * E:(n + 1) * E]
        n3 = 12 if n == 0 else 1
        out += [f[:256 * 144].view(256, 144)[:, :137], f[102400 + 4096:102400 + 4096 + 256],
                f[36864:36864 + 65536].view(256, 256), f[102400 + 4096 + 256:102400 + 4096 + 512],
                f[102400:102400 + 4096].view(16, 256)[:n3], f[102400 + 4096 + 512:102400 + 4096 + 512 + 16][:n3]]
    return out


@pytest.mark.parametrize("B", [8192, 4099, 1])
def test_train_step_equals_autograd_plus_torch_adam(B):
    """pnr_mlp_train_step over several updates on changing minibatches.  (a) Its gradient — the flat bucket of the fused
    forward + loss + backward kernel and the weight-gradient kernel — against autograd through the SEPARATE forward / loss /
    backward kernels, parameter by parameter; (b) its optimiser arithmetic (update count on the device, bias correction,
    bf16 repacking) against torch.optim.Adam fed with the same gradients; (c) the fused reduce + Adam form against the
    flat-bucket form (reduce -> [all-reduce] -> pnr_mlp_adam): bit for bit."""
    import copy
    from pioneer_amd.mlp import HipMLP
    R, lr = 20000, 1e-3           # B: full tiles | a ragged last tile and slice (4099 = 64 * 64 + 3) | a single sample
    model, mlp, obs, _, filt = make(B, seed=21, rows=R, with_filter=True)
    dev = obs.device
    rec = _record(R, dev)
    klc = torch.tensor(0.2, device=dev); entc = torch.tensor(0.01, device=dev)
    model_a = copy.deepcopy(model); mlp_a = HipMLP(model_a, B, dev)          # autograd through the separate kernels
    model_t = copy.deepcopy(model)                                          # torch Adam on the HIP gradients
    model_f = copy.deepcopy(model); mlp_f = HipMLP(model_f, B, dev)          # the flat-bucket form
    params_t = [p for net in (model_t.policy, model_t.value) for l in net if isinstance(l, torch.nn.Linear) for p in (l.weight, l.bias)]
    assert [tuple(p.shape) for p in params_t] == [tuple(p.shape) for p in mlp.params]
    opt = torch.optim.Adam(params_t, lr=lr)
    means = torch.zeros(6, 8, device=dev); means_f = torch.zeros(6, 8, device=dev)
    flat = torch.zeros(int(mlp.lib.pnr_mlp_grad_floats()), device=dev)
    mlp.pack(); mlp_f.pack()
    g = torch.Generator(device=dev).manual_seed(1)
    for it in range(6):
        idx = torch.randperm(R, generator=g, device=dev)[:B].contiguous()
        if it == 0:
            m = mlp_a.policy_loss(obs, idx, filt, dict(rec, obs=obs), klc, entc, 0.3, 10.0, 1.0)
            m[4].backward()
        mlp.train_step(obs, idx, filt, rec, klc, entc, 0.3, 10.0, 1.0, means[it], lr)
        mlp_f.train_step(obs, idx, filt, rec, klc, entc, 0.3, 10.0, 1.0, means_f[it], lr, flat_grad=flat)
        if it == 0:
            for gk, prm in zip(_unpack_flat(flat), mlp_a.params):
                assert rel(gk, prm.grad) < 2e-3, (tuple(prm.shape), rel(gk, prm.grad))
            assert float(flat[:256 * 144].view(256, 144)[:, 137:].abs().max()) == 0.0       # the padded input columns
            # (the fused kernel sums the loss per 64-sample tile and net, the separate loss kernel per 256 samples)
            assert torch.allclose(means[0, :5], m.detach()[:5], rtol=1e-4, atol=1e-6)
        for prm, gk in zip(params_t, _unpack_flat(flat)):
            prm.grad = gk.clone()
        opt.step()
        mlp_f.adam(flat, 1.0, lr)
        assert torch.equal(means[it], means_f[it])
        for a, b in zip(mlp_f.params, params_t):                  # same gradients: the two Adams agree to rounding
            assert float((a - b).abs().max()) <= 2e-3 * lr, (it, float((a - b).abs().max()))
    assert float(mlp.adam_state()[2]) == 6.0
    for a, c in zip(mlp.params, mlp_f.params):
        assert torch.equal(a, c)                                  # fused == flat-bucket form
    assert float((mlp.params[2] - model.policy[2].weight).abs().max()) == 0.0     # the module's own tensors were updated
    w_now = mlp.wpack.clone(); b_now = mlp.bias.clone()
    mlp.pack()
    assert torch.equal(w_now, mlp.wpack) and torch.equal(b_now, mlp.bias)         # the refreshed bf16 copies are exact


def test_adam_refuses_a_gradient_that_is_not_16_byte_aligned():
    """The optimiser kernel moves its operands four floats at a time (include/pioneer_amd.h, pnr_mlp_step): a misaligned
    bucket is an argument error, not a fault."""
    from pioneer_amd._lib import PnrError
    model, mlp, obs, _, filt = make(64, seed=3, rows=64, with_filter=False)
    mlp.pack()
    n = int(mlp.lib.pnr_mlp_grad_floats())
    store = torch.zeros(n + 4, device=obs.device)
    before = [p.detach().clone() for p in mlp.params]
    with pytest.raises(PnrError, match="16-byte aligned"):
        mlp.adam(store[1:n + 1], 1.0, 1e-3)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(before, mlp.params))
    mlp.adam_state()[2].fill_(1.0)               # (the update count the fused kernel would have written)
    mlp.adam(store[4:n + 4], 1.0, 1e-3)          # aligned view of the same storage: accepted (zero gradient: parameters unchanged)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(before, mlp.params))


@pytest.mark.parametrize("B", [8192, 4099, 16421])
def test_pre_gathered_epoch_equals_the_in_kernel_gather(B):
    """pnr_mlp_gather + train_step(xs_in=...) against train_step(obs, idx, filt): the same filter arithmetic and rounding,
    the same record rows, so gradients, loss means and updated weights are bit-identical — on a minibatch that is a slice
    of a longer gathered epoch (pointer offsets into the gathered arrays)."""
    import copy
    from pioneer_amd.mlp import HipMLP
    R, lr = 20000, 1e-3
    model, mlp, obs, _, filt = make(B, seed=33, rows=R, with_filter=True)
    dev = obs.device
    rec = _record(R, dev)
    klc = torch.tensor(0.2, device=dev); entc = torch.tensor(0.01, device=dev)
    model_g = copy.deepcopy(model); mlp_g = HipMLP(model_g, B, dev)
    mlp.pack(); mlp_g.pack()
    g = torch.Generator(device=dev).manual_seed(5)
    perm = torch.randperm(R, generator=g, device=dev)
    gathered = mlp_g.gather_epoch(obs, perm, filt, rec)
    assert torch.equal(gathered["actions"], rec["actions"][perm]) and torch.equal(gathered["adv"], rec["adv"][perm])
    # the same through the packed 24-float rows (pnr_ppo_pack_record), also with the advantages standardised on the way
    snap = {k: v.clone() for k, v in gathered.items()}
    via_rows = mlp_g.gather_epoch(obs, perm, filt, None, rec_rows=mlp_g.pack_record(rec))
    assert all(torch.equal(via_rows[k], snap[k]) for k in snap)
    # ... and from the sampler's saved net inputs instead of the float32 observations
    xs_all = torch.zeros(R, 144, dtype=torch.bfloat16, device=dev)
    xs_all[:, :137] = torch.clamp((obs - filt[0]) * filt[1], min=filt[2], max=filt[3]).to(torch.bfloat16)
    via_xs = mlp_g.gather_epoch(None, perm, None, None, rec_rows=mlp_g.pack_record(rec), xs_rows=xs_all)
    assert all(torch.equal(via_xs[k], snap[k]) for k in snap)
    mu = torch.tensor([0.3], device=dev); den = torch.tensor([1.7], device=dev)
    std = mlp_g.gather_epoch(obs, perm, filt, None, rec_rows=mlp_g.pack_record(rec, mu, den))
    assert torch.equal(std["adv"], ((rec["adv"] - mu) / den)[perm]) and torch.equal(std["vtarg"], rec["vtarg"][perm])
    gathered = mlp_g.gather_epoch(obs, perm, filt, rec)
    x = torch.clamp((obs[perm] - filt[0]) * filt[1], min=filt[2], max=filt[3]).to(torch.bfloat16)
    assert torch.equal(gathered["xs"][:, :137], x) and float(gathered["xs"][:, 137:].float().abs().max()) == 0.0
    means = torch.zeros(2, 8, device=dev); means_g = torch.zeros(2, 8, device=dev)
    flat = torch.zeros(int(mlp.lib.pnr_mlp_grad_floats()), device=dev); flat_g = torch.zeros_like(flat)
    for it, s0 in enumerate((0, R - B)):                       # the first and the last B rows of the epoch
        idx = perm[s0:s0 + B].contiguous()
        mlp.train_step(obs, idx, filt, rec, klc, entc, 0.3, 10.0, 1.0, means[it], lr, flat_grad=flat)
        mlp_g.train_step(None, None, None, {k: gathered[k][s0:s0 + B] for k in mlp_g.REC_KEYS}, klc, entc, 0.3, 10.0, 1.0, means_g[it], lr,
                         flat_grad=flat_g, xs_in=gathered["xs"][s0:s0 + B])
        assert torch.equal(flat, flat_g) and torch.equal(means[it], means_g[it])
        mlp.adam(flat, 1.0, lr); mlp_g.adam(flat_g, 1.0, lr)
    for a, b in zip(mlp.params, mlp_g.params):
        assert torch.equal(a, b)


# ---- float32-accurate operands: every MFMA operand as split 16-bit planes (include/pioneer_amd.h, pnr_mlp_pack) -----------------------
# The reference's learner is float32 torch (pioneer_knm_train.py:47): tolerances against plain float32 torch autograd, written here
# (the r04 bounds, unchanged):
#   planes = 2 ("f32", r05): two scaled fp16 planes, 22 significant bits, three MFMAs per product
#   planes = 3 ("bf16x3", r04): three bf16 planes, 24 bits, six MFMAs per product
#   both: heads relative L2 <= 1e-5 vs float64 AND within 4x of torch float32's own distance to float64 (~4e-7);
#         gradients <= 1e-4 relative L2 vs float32 autograd (measured: tools/split_accuracy.py, profiles/r05_*_accuracy_by_precision.jsonl)
SPLIT_TOL = {3: (1e-5, 1e-4), 2: (1e-5, 1e-4)}


def _f32_reference(model, x):
    with torch.no_grad():
        return model.policy(x), model.value(x)


@pytest.mark.parametrize("planes", [3, 2])
@pytest.mark.parametrize("B,rows,with_filter", [(4099, 9000, True), (32768, None, False), (1, None, True)])
def test_split_operand_forward_matches_float32_torch(planes, B, rows, with_filter):
    from pioneer_amd.mlp import HipMLP
    model, _, obs, idx, filt = make(B, seed=31, rows=rows, with_filter=with_filter)
    mlp = HipMLP(model, B, obs.device, planes=planes)
    mlp.pack()
    head = mlp.forward_nograd(obs, idx, filt)
    x = net_input(obs, idx, filt)
    ref_p, ref_v = _f32_reference(model, x)
    ref64_p, ref64_v = model.double().policy(x.double()), model.double().value(x.double())
    model.float()
    tol = SPLIT_TOL[planes][0]
    if B == 1:      # one value-head number (here -0.008, a sum of terms ~0.1): relative error of a single cancelling scalar says nothing
        assert float((head[1, :, :1].double() - ref64_v).abs().max()) <= tol * 0.5
    else:
        assert rel(head[1, :, :1], ref64_v) <= tol, rel(head[1, :, :1], ref64_v)
    assert rel(head[0, :, :12], ref64_p) <= tol, rel(head[0, :, :12], ref64_p)
    # as close to the float64 truth as torch's own float32 forward is (both a few 1e-7): float32-equivalent
    assert rel(head[0, :, :12], ref64_p) <= max(4.0 * rel(ref_p, ref64_p), 1e-6), (rel(head[0, :, :12], ref64_p), rel(ref_p, ref64_p))
    # the bf16 path on the same inputs, for scale
    mlp1 = HipMLP(model, B, obs.device); mlp1.pack()
    assert rel(mlp1.forward_nograd(obs, idx, filt)[0, :, :12], ref64_p) > 50 * rel(head[0, :, :12], ref64_p)


@pytest.mark.parametrize("planes", [3, 2])
@pytest.mark.parametrize("B", [8192, 4099])
def test_split_operand_train_step_matches_float32_autograd_and_adam(planes, B):
    """pnr_mlp_train_step with split operands on pre-gathered input planes: (a) its gradient (the flat bucket) against float32 torch
    autograd of the PPO loss (PPOLearner.loss: the torch formulation), parameter by parameter; (b) the loss means; (c) the update
    against torch.optim.Adam on the autograd gradients; (d) the fused reduce + Adam form against the flat-bucket form, bit for bit;
    (e) every plane of the repacked weights against a fresh pack."""
    import copy
    from pioneer_amd.mlp import HipMLP
    from pioneer_amd.ppo import PPOConfig, PPOLearner
    R, lr = 12000, 1e-3
    model, _, obs, _, filt = make(B, seed=23, rows=R, with_filter=True)
    dev = obs.device
    rec = _record(R, dev)
    klc = torch.tensor(0.2, device=dev); entc = torch.tensor(0.01, device=dev)
    mlp = HipMLP(model, B, dev, planes=planes)
    model_f = copy.deepcopy(model); mlp_f = HipMLP(model_f, B, dev, planes=planes)
    # the torch formulation on a copy of the weights
    cfg = PPOConfig(lr=lr, hip_kernels=False, kl_coeff=0.2, vf_clip_param=10.0, clip_param=0.3, vf_loss_coeff=1.0)
    L = PPOLearner(cfg, dev)
    L.model.load_state_dict(model.state_dict())
    L.kl_coeff = 0.2
    params_l = [p for net in (L.model.policy, L.model.value) for l in net if isinstance(l, torch.nn.Linear) for p in (l.weight, l.bias)]
    model_t = copy.deepcopy(model)                               # torch.optim.Adam fed with the kernels' gradients
    params_t = [p for net in (model_t.policy, model_t.value) for l in net if isinstance(l, torch.nn.Linear) for p in (l.weight, l.bias)]
    opt = torch.optim.Adam(params_t, lr=lr)
    means = torch.zeros(4, 8, device=dev); means_f = torch.zeros(4, 8, device=dev)
    flat = torch.zeros(int(mlp.lib.pnr_mlp_grad_floats()), device=dev)
    mlp.pack(); mlp_f.pack()
    g = torch.Generator(device=dev).manual_seed(2)
    gtol = SPLIT_TOL[planes][1]
    for it in range(4):
        perm = torch.randperm(R, generator=g, device=dev).contiguous()
        ga = mlp.gather_epoch(obs, perm, filt, rec)
        gb = mlp_f.gather_epoch(obs, perm, filt, rec)
        s0 = 64 * 7                                               # a minibatch inside the epoch's rows: plane stride R * 144
        mb = {k: ga[k][s0:s0 + B] for k in mlp.REC_KEYS}
        mlp.train_step(None, None, None, mb, klc, entc, 0.3, 10.0, 1.0, means[it], lr, xs_in=ga["xs"][:, s0:s0 + B])
        mlp_f.train_step(None, None, None, {k: gb[k][s0:s0 + B] for k in mlp.REC_KEYS}, klc, entc, 0.3, 10.0, 1.0, means_f[it], lr,
                         flat_grad=flat, xs_in=gb["xs"][:, s0:s0 + B])
        # float32 autograd on the same minibatch
        rows_ = perm[s0:s0 + B]
        x = net_input(obs, rows_, filt)
        L._ent_c.fill_(0.01); L._kl_c.fill_(0.2)
        batch = {"obs": x, "actions": rec["actions"][rows_], "mean": rec["mean"][rows_], "log_std": rec["log_std"][rows_],
                 "logp": rec["logp"][rows_], "adv": rec["adv"][rows_], "vtarg": rec["vtarg"][rows_], "values": rec["values"][rows_]}
        for prm in params_l:
            prm.grad = None
        loss, info = L.loss(batch)
        loss.backward()
        for gk, prm, nm in zip(_unpack_flat(flat), params_l, NAMES):
            assert rel(gk, prm.grad) <= gtol, (it, nm, rel(gk, prm.grad))
        assert abs(float(means[it, 4]) - float(loss)) <= 2e-5 * max(1.0, abs(float(loss))), (float(means[it, 4]), float(loss))
        for prm, gk in zip(params_t, _unpack_flat(flat)):
            prm.grad = gk.clone()
        opt.step()
        mlp_f.adam(flat, 1.0, lr)
        assert torch.equal(means[it], means_f[it])
        for a, b in zip(mlp_f.params, params_t):                  # same gradients: the two Adams agree to rounding
            assert float((a - b).abs().max()) <= 2e-3 * lr, (it, float((a - b).abs().max()))
        with torch.no_grad():                                     # the autograd reference follows the kernels' weights
            for a, b in zip(params_l, mlp_f.params):
                a.copy_(b)
    for a, c in zip(mlp.params, mlp_f.params):
        assert torch.equal(a, c)                                  # fused reduce + Adam == flat-bucket form
    w_now = mlp.wpack.clone(); b_now = mlp.bias.clone()
    mlp.pack()
    assert torch.equal(w_now, mlp.wpack) and torch.equal(b_now, mlp.bias)         # every plane of the refreshed weights is exact
    # the planes add up to the float32 master weights (3 bf16 planes: exactly; 2 fp16 planes of 256 w: to 2^-22)
    n = int(mlp.lib.pnr_mlp_pack_elems())
    if planes == 2:
        tot = mlp.wpack.view(torch.float16).view(planes, n).float().sum(0) / 256.0
    else:
        tot = mlp.wpack.view(planes, n).float().sum(0)
    one = HipMLP(model, B, dev); one.pack()
    assert float((tot - one.wpack.float()).abs().max()) <= 2.0 ** -8 * float(one.wpack.float().abs().max())
    w2 = mlp.params[2].detach()                                   # .. checked where the packing is the identity map's inverse: max |error| of W2
    assert float(tot.abs().max()) >= 0.9 * float(w2.abs().max())


def test_fp16_planes_saturate_instead_of_overflowing():
    """planes = 2 carries every operand as scaled fp16 planes, whose largest finite value is 65 504 (include/pioneer_amd.h, `planes`):
    net inputs beyond 4 094 (unfiltered observations: the env does not clip actions), weights beyond 255 and per-sample gradients beyond
    16 384 / batch are CLAMPED at the conversion — never inf, never NaN: the update of such a batch stays finite, the samples that are in
    range are unaffected, and the in-range answer agrees with float32 autograd as always."""
    from pioneer_amd.mlp import HipMLP
    B = 4096
    model, _, obs, _, _ = make(B, seed=41)
    dev = obs.device
    rec = _record(B, dev)
    klc = torch.tensor(0.2, device=dev); entc = torch.tensor(0.01, device=dev)

    def grad_of(o, r):
        mlp = HipMLP(model, B, dev, planes=2)
        mlp.pack()
        g = mlp.gather_epoch(o, torch.arange(B, device=dev), None, r)
        flat = torch.zeros(int(mlp.lib.pnr_mlp_grad_floats()), device=dev)
        means = torch.zeros(8, device=dev)
        mlp.train_step(None, None, None, {k: g[k] for k in mlp.REC_KEYS}, klc, entc, 0.3, 10.0, 1.0, means, 1e-3, flat_grad=flat, xs_in=g["xs"])
        head = mlp.forward_nograd(o)
        torch.cuda.synchronize()
        return flat, means, head

    clean, _, head_clean = grad_of(obs, rec)
    wild_obs = obs.clone(); wild_obs[7, :] = 3.0e7; wild_obs[8, 5] = -1.0e30; wild_obs[9, 0] = float(2 ** 15)      # far beyond the fp16 range
    wild_rec = {k: v.clone() for k, v in rec.items()}
    wild_rec["adv"][11] = 1.0e9; wild_rec["adv"][12] = -1.0e12; wild_rec["vtarg"][13] = 1.0e20                     # per-sample gradients beyond it
    flat, means, head = grad_of(wild_obs, wild_rec)
    assert torch.isfinite(flat).all() and torch.isfinite(head).all()
    keep = torch.ones(B, dtype=torch.bool, device=dev); keep[7:10] = False
    assert torch.equal(head[:, keep], head_clean[:, keep]), "samples inside the range must not notice their neighbours"
    # the clamp is where the header says: an input of 2^15 enters the first layer as 65504 / 16
    x = wild_obs[9:10].clone(); x[0, 0] = 65504.0 / 16.0
    mlp = HipMLP(model, B, dev, planes=2); mlp.pack()
    assert torch.equal(mlp.forward_nograd(torch.cat([x, obs[:1]]))[:, 0], head[:, 9])
    assert float((flat - clean).abs().max()) > 0 and float(flat.abs().max()) < 1e6
