"""Unit parity of every HIP kernel behind the C ABI vs the CPU oracle (float64 where cheap).
Integer/index results are bit-exact; float tolerances are written next to each check."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_cpu
from tests.helpers import describe, rel_l2
from versband_amd import _lib as L
from versband_amd import pack, prng

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return L.load()


_KEEP = []


def dev(t):
    """upload and KEEP the tensor alive: a temporary would be freed (and its block reused by the next
    upload) before the asynchronous kernel that reads it has run."""
    d = t.cuda().contiguous()
    _KEEP.append(d)
    if len(_KEEP) > 64:
        torch.cuda.synchronize()
        del _KEEP[:32]
    return d


def rnd(shape, name, scale=1.0):
    n = int(np.prod(shape))
    return torch.from_numpy(prng.normal(prng.key_seed(7, name), n).reshape(shape)) * scale


def sync():
    torch.cuda.synchronize()


# ---------------------------------------------------------------- GEMM ------
@pytest.mark.parametrize("M,N,K", [(200, 132, 96), (1504, 768, 768), (129, 2304, 768), (77, 64, 1024), (300, 192, 512)])
@pytest.mark.parametrize("npl", [1, 2])
def test_gemm_bf16(lib, M, N, K, npl):
    A, B, bias = rnd((M, K), "gA"), rnd((N, K), "gB", 0.05), rnd((N,), "gbias")
    Ap, Bp = dev(pack.to_planes(A, npl)), dev(pack.to_planes(B, npl))
    Cd = torch.full((M, N), float("nan"), device="cuda")
    L.check(lib.vb_gemm_bf16(L.ptr(Ap), L.ptr(Bp), L.ptr(dev(bias)), M, N, K, npl, L.ptr(Cd), L.stream_ptr()), "gemm")
    sync()
    if npl == 1:   # exact products of the bf16-rounded operands, fp32 accumulate
        ref = Ap[0].float().double().cpu() @ Bp[0].float().double().cpu().T + bias.double()
        tol = 2e-6
    else:          # split precision: fp32-class
        ref = A.double() @ B.double().T + bias.double()
        tol = 3e-5
    assert rel_l2(Cd, ref) < tol, describe(f"gemm {M}x{N}x{K} np={npl}", Cd, ref)


# ---------------------------------------------------------------- conv ------
CONV_CASES = [  # B, Ci, T, Co, k, dil, in_act, res
    (2, 20, 50, 1536, 5, 1, 0, False),
    (1, 32, 300, 32, 11, 5, 1, True),
    (2, 80, 37, 80, 3, 1, 0, False),
    (1, 64, 200, 64, 7, 3, 1, True),
    (1, 32, 500, 1, 7, 1, 1, False),
    (2, 1536, 40, 768, 3, 1, 0, False),
    (1, 256, 130, 256, 3, 3, 1, True),
]


@pytest.mark.parametrize("B,Ci,T,Co,k,dil,act,res", CONV_CASES)
@pytest.mark.parametrize("split", [False, True])
def test_conv1d_f32(lib, B, Ci, T, Co, k, dil, act, res, split):
    x, w, b = rnd((B, Ci, T), "cx"), rnd((Co, Ci, k), "cw", 1.0 / (Ci * k) ** 0.5), rnd((Co,), "cb")
    r = rnd((B, Co, T), "cr") if res else None
    pad = (k - 1) * dil // 2
    out = torch.full((B, Co, T), float("nan"), device="cuda")
    wx3, cip = pack.pack_conv_x3(pack.pack_conv(w))
    L.check(lib.vb_conv1d_f32(L.ptr(dev(x)), L.ptr(dev(pack.pack_conv(w))), L.ptr(dev(b)), B, Ci, T, Co, k, dil, pad, 1, 0, 0, T,
                              act, 0.1, L.ptr(dev(r)) if res else None, L.ptr(out), L.ptr(dev(wx3)) if split else None, cip,
                              L.stream_ptr()), "conv")
    sync()
    xin = F.leaky_relu(x.double(), 0.1) if act else x.double()
    ref = F.conv1d(xin, w.double(), b.double(), dilation=dil, padding=pad)
    if res:
        ref = ref + r.double()
    # exact-f32 MFMA: fp32 roundoff only; split-bf16: 2^-17 operand error
    assert rel_l2(out, ref) < (2e-5 if split else 2e-6), describe("conv1d", out, ref)


@pytest.mark.parametrize("B,Ci,T,Co,k,dil", [(2, 128, 520, 128, 3, 1), (1, 256, 388, 256, 7, 3), (2, 128, 300, 256, 11, 5), (3, 128, 752, 1536, 3, 1)])
def test_fp32_dma_conv_unrolled_taps_equal_the_runtime_tap_loop(lib, monkeypatch, B, Ci, T, Co, k, dil):
    """conv1d_f32g_kernel with the tap count as a template parameter (3 / 7 / 11: waits and addresses of a chunk's steps are immediates)
    against the same kernel's runtime-tap loop (VB_CONV_F32_RT_TAPS=1): same tiles in the same order - the same bits."""
    x, w, b = dev(rnd((B, Ci, T), "ux")), rnd((Co, Ci, k), "uw", 1.0 / (Ci * k) ** 0.5), dev(rnd((Co,), "ub"))
    r = dev(rnd((B, Co, T), "ur"))
    pad = (k - 1) * dil // 2
    wpk = dev(pack.pack_conv(w))
    new = _conv_f32(lib, x, wpk, b, r, B, Ci, T, Co, k, dil, pad, 1)
    monkeypatch.setenv("VB_CONV_F32_RT_TAPS", "1")
    lib.vb_tune_reload()
    old = _conv_f32(lib, x, wpk, b, r, B, Ci, T, Co, k, dil, pad, 1)
    monkeypatch.delenv("VB_CONV_F32_RT_TAPS")
    lib.vb_tune_reload()
    assert torch.isfinite(new).all() and torch.equal(new, old), f"unrolled vs runtime taps differ by {float((new - old).abs().max()):.3e}"


@pytest.mark.parametrize("B,Ci,T,Co,k,u", [(2, 512, 24, 256, 16, 8), (1, 256, 33, 128, 15, 5), (1, 128, 20, 64, 11, 5),
                                          (2, 64, 50, 32, 4, 2), (1, 128, 19, 64, 8, 4)])
@pytest.mark.parametrize("split", [False, True])
def test_conv_transpose1d_f32(lib, B, Ci, T, Co, k, u, split):
    x, w, b = rnd((B, Ci, T), "tx"), rnd((Ci, Co, k), "tw", (u / (Ci * k)) ** 0.5), rnd((Co,), "tb")
    p = (k - u) // 2
    ref = F.conv_transpose1d(F.leaky_relu(x.double(), 0.1), w.double(), b.double(), stride=u, padding=p)
    T_out = ref.shape[-1]
    out = torch.full((B, Co, T_out), float("nan"), device="cuda")
    wx3, cip = pack.pack_conv_x3(pack.pack_conv_transpose(w, u))
    L.check(lib.vb_conv1d_f32(L.ptr(dev(x)), L.ptr(dev(pack.pack_conv_transpose(w, u))), L.ptr(dev(b)), B, Ci, T, Co, 0, 1, 0, u, p,
                              k, T_out, 1, 0.1, None, L.ptr(out), L.ptr(dev(wx3)) if split else None, cip, L.stream_ptr()), "convT")
    sync()
    assert rel_l2(out, ref) < (2e-5 if split else 2e-6), describe("conv_transpose1d", out, ref)


# ---------------------------------------------------------------- attention ------
@pytest.mark.parametrize("B,T,Lc,H,cross,self_", [(2, 200, 80, 8, True, True), (1, 64, 80, 2, True, False), (1, 752, 80, 1, True, True),
                                                 (2, 130, 16, 3, False, True)])
@pytest.mark.parametrize("npl", [1, 2])
def test_attention(lib, B, T, Lc, H, cross, self_, npl):
    hd = 96
    q, k, v = rnd((B, T, H, hd), "aq"), rnd((B, T, H, hd), "ak"), rnd((B, T, H, hd), "av")
    ky, vy = rnd((B, Lc, H, hd), "aky"), rnd((B, Lc, H, hd), "avy")
    cw = rnd((H,), "acw")
    Tpad, Lpad = (T + 63) // 64 * 64, (Lc + 63) // 64 * 64

    def vt_layout(x, S, Spad):   # [B,S,H,hd] -> [B,H,hd,Spad] zero padded
        o = torch.zeros(B, H, hd, Spad)
        o[..., :S] = x.permute(0, 2, 3, 1)
        return o
    qp, kp, vtp = dev(pack.to_planes(q, npl)), dev(pack.to_planes(k, npl)), dev(pack.to_planes(vt_layout(v, T, Tpad), npl))
    kyp, vytp = dev(pack.to_planes(ky, npl)), dev(pack.to_planes(vt_layout(vy, Lc, Lpad), npl))
    out = torch.zeros(npl, B, T, H, hd, dtype=torch.bfloat16, device="cuda")
    L.check(lib.vb_attention(L.ptr(qp), L.ptr(kp) if self_ else None, L.ptr(vtp) if self_ else None, L.ptr(kyp) if cross else None,
                             L.ptr(vytp) if cross else None, L.ptr(dev(cw)) if cross else None, B, T, Tpad, Lc, Lpad, H, hd, npl,
                             L.ptr(out), L.stream_ptr()), "attention")
    sync()
    got = pack.planes_to_float(out.cpu())
    src = (lambda t, pl: pack.planes_to_float(pl.cpu()).reshape(t.shape)) if npl == 1 else (lambda t, pl: t)
    qq, kk = src(q, qp).double(), src(k, kp).double()
    vv = v.to(torch.bfloat16).double() if npl == 1 else v.double()
    kyy, vyy = src(ky, kyp).double(), (vy.to(torch.bfloat16).double() if npl == 1 else vy.double())
    ref = torch.zeros(B, H, T, hd, dtype=torch.float64)
    if self_:
        ref = ref + ref_cpu.sdpa(qq.permute(0, 2, 1, 3), kk.permute(0, 2, 1, 3), vv.permute(0, 2, 1, 3))
    if cross:
        ref = ref + ref_cpu.sdpa(qq.permute(0, 2, 1, 3), kyy.permute(0, 2, 1, 3), vyy.permute(0, 2, 1, 3)) * cw.double().view(1, H, 1, 1)
    ref = ref.permute(0, 2, 1, 3)
    # np=1: P and the output are rounded to bf16 (2^-9 relative each); np=2: fp32-class
    tol = 6e-3 if npl == 1 else 4e-5
    assert rel_l2(got, ref) < tol, describe(f"attention np={npl}", got, ref)


@pytest.mark.parametrize("npl", [1, 2])
def test_attention_deferred_rescale_on_spiked_keys(lib, npl, monkeypatch):
    """The kernel moves its running row maximum only when a row outgrows it by 2^thr (VB_ATTN_DEFER, log2 units).  Keys that
    dwarf everything seen before - late in the sequence, for a few rows only, in the middle of a 64-key tile - force that
    rare branch; thr = 0 (exact running maximum), the default and "never again after the first tile" (thr = 1e9) must all
    agree with the float64 reference, and the default with thr = 0 to rounding."""
    B, T, H, hd, Lc = 1, 400, 2, 96, 20
    q, k, v = rnd((B, T, H, hd), "dq"), rnd((B, T, H, hd), "dk"), rnd((B, T, H, hd), "dv")
    ky, vy, cw = rnd((B, Lc, H, hd), "dky"), rnd((B, Lc, H, hd), "dvy"), rnd((H,), "dcw")
    for row, key, gain in ((7, 150, 30.0), (7, 333, 60.0), (200, 390, 45.0), (399, 70, 25.0)):
        k[0, key] = q[0, row] * gain / q[0, row].pow(2).sum(-1, keepdim=True).sqrt()     # q_row . k_key ~ gain * |q_row|: scores far above the rest
    Tpad, Lpad = (T + 63) // 64 * 64, 64

    def vt_layout(x, S, Spad):
        o = torch.zeros(B, H, hd, Spad)
        o[..., :S] = x.permute(0, 2, 3, 1)
        return o
    qp, kp, vtp = dev(pack.to_planes(q, npl)), dev(pack.to_planes(k, npl)), dev(pack.to_planes(vt_layout(v, T, Tpad), npl))
    kyp, vytp = dev(pack.to_planes(ky, npl)), dev(pack.to_planes(vt_layout(vy, Lc, Lpad), npl))
    src = (lambda t, pl: pack.planes_to_float(pl.cpu()).reshape(t.shape)) if npl == 1 else (lambda t, pl: t)
    qq, kk, kyy = src(q, qp).double(), src(k, kp).double(), src(ky, kyp).double()
    vv, vyy = (v.to(torch.bfloat16).double(), vy.to(torch.bfloat16).double()) if npl == 1 else (v.double(), vy.double())
    ref = ref_cpu.sdpa(qq.permute(0, 2, 1, 3), kk.permute(0, 2, 1, 3), vv.permute(0, 2, 1, 3)) \
        + ref_cpu.sdpa(qq.permute(0, 2, 1, 3), kyy.permute(0, 2, 1, 3), vyy.permute(0, 2, 1, 3)) * cw.double().view(1, H, 1, 1)
    ref = ref.permute(0, 2, 1, 3)
    outs = {}
    for thr in ("0", None, "1e9"):
        if thr is None:
            monkeypatch.delenv("VB_ATTN_DEFER", raising=False)
        else:
            monkeypatch.setenv("VB_ATTN_DEFER", thr)
        lib.vb_tune_reload()
        out = torch.zeros(npl, B, T, H, hd, dtype=torch.bfloat16, device="cuda")
        L.check(lib.vb_attention(L.ptr(qp), L.ptr(kp), L.ptr(vtp), L.ptr(kyp), L.ptr(vytp), L.ptr(dev(cw)), B, T, Tpad, Lc, Lpad, H, hd, npl,
                                 L.ptr(out), L.stream_ptr()), "attention")
        sync()
        outs[thr] = pack.planes_to_float(out.cpu())
        assert torch.isfinite(outs[thr]).all()
        tol = 6e-3 if npl == 1 else 4e-5
        if thr != "1e9":         # (never rescaling after the first tile loses the small rows' precision by construction: finite, not accurate)
            assert rel_l2(outs[thr], ref) < tol, describe(f"attention defer={thr} np={npl}", outs[thr], ref)
            # the spiked rows themselves (a relative norm over the whole tensor would hide four bad rows)
            for row in (7, 200, 399):
                assert rel_l2(outs[thr][0, row], ref[0, row]) < tol, describe(f"row {row} defer={thr}", outs[thr][0, row], ref[0, row])
    monkeypatch.delenv("VB_ATTN_DEFER", raising=False)
    lib.vb_tune_reload()
    assert rel_l2(outs[None], outs["0"]) < (4e-3 if npl == 1 else 2e-5)


# ---------------------------------------------------------------- small kernels ------
@pytest.mark.parametrize("npl", [1, 2])
def test_rmsnorm_modulate(lib, npl):
    B, T, D = 3, 37, 768
    h, w = rnd((B * T, D), "rh", 2.0), rnd((D,), "rw") * 0.2 + 1
    mod = rnd((B, 6 * D), "rmod", 0.3)
    out = torch.zeros(npl, B * T, D, dtype=torch.bfloat16, device="cuda")
    md = dev(mod)
    L.check(lib.vb_rmsnorm_modulate(L.ptr(dev(h)), L.ptr(dev(w)), C.c_void_p(md.data_ptr()), C.c_void_p(md.data_ptr() + 4 * D), 6 * D,
                                    B * T, D, T, 1e-5, L.ptr(out), npl, L.stream_ptr()), "rmsnorm")
    sync()
    ref = ref_cpu.modulate(ref_cpu.rmsnorm(h.view(B, T, D), w, 1e-5), mod[:, :D], mod[:, D:2 * D]).reshape(B * T, D)
    got = pack.planes_to_float(out.cpu())
    tol = 4e-3 if npl == 1 else 2e-5
    assert rel_l2(got, ref) < tol, describe("rmsnorm_modulate", got, ref)


@pytest.mark.parametrize("E", [4, 8])
def test_router_top1_bit_exact(lib, E):
    N = 5000
    logits, gum = rnd((N, E), "rl"), ref_cpu.gumbel_from_exponential(torch.from_numpy(prng.exponential(11, N * E).reshape(N, E)))
    logits[:50] = torch.round(logits[:50])         # force exact ties: the first maximum must win
    gum[:50] = 0.0
    idx = torch.full((N,), -1, dtype=torch.int32, device="cuda")
    L.check(lib.vb_router_top1(L.ptr(dev(logits)), L.ptr(dev(gum)), N, E, L.ptr(idx), L.stream_ptr()), "router_top1")
    sync()
    ref, _ = ref_cpu.router_top1(logits, gum, 2.0)
    assert torch.equal(idx.cpu().long(), ref), f"{int((idx.cpu().long() != ref).sum())} of {N} routing indices differ"


@pytest.mark.parametrize("N,E", [(1, 4), (1000, 4), (1504, 4), (4096, 4), (4097, 4), (12032, 4), (3000, 8), (6016, 8)])
def test_route_bucket(lib, N, E):
    ic = torch.from_numpy(prng.randint(3, N, 0, E)).int()
    ia = torch.from_numpy(prng.randint(4, N, 0, E)).int()
    if N > 100:
        ic[ic == 2] = 1                                # an empty group
    off = torch.full((2 * E + 1,), -1, dtype=torch.int32, device="cuda")
    perm = torch.full((2 * N + lib.vb_route_bucket_scratch_ints(N, E),), -1, dtype=torch.int32, device="cuda")
    L.check(lib.vb_route_bucket(L.ptr(dev(ic)), L.ptr(dev(ia)), N, E, L.ptr(off), L.ptr(perm), L.stream_ptr()), "bucket")
    sync()
    off, perm = off.cpu(), perm.cpu()[:2 * N]
    cnt = torch.cat([torch.bincount(ic.long(), minlength=E), torch.bincount(ia.long(), minlength=E)])
    assert torch.equal(off, torch.cat([torch.zeros(1, dtype=torch.long), cnt.cumsum(0)]).int())
    for g in range(2 * E):
        sel = perm[off[g]:off[g + 1]].long()
        src = ic if g < E else ia
        want = (src == (g % E)).nonzero().squeeze(1)
        assert torch.equal(sel, want), f"group {g}: stable order violated"      # stable: tokens in ascending order


@pytest.mark.parametrize("N,E", [(5000, 4), (12032, 4), (6016, 2), (777, 3), (1504, 4), (4096, 4), (1, 2)])
def test_route_bucket_pairs(lib, N, E):
    """pair mode: one rank per token by (caption, acoustic) expert pair; caption slots = pair slots (caption-major), acoustic slots =
    the same buckets acoustic-major; group_off / perm stay a valid bucketing by expert, tokens ascending inside a PAIR bucket"""
    ic = torch.from_numpy(prng.randint(13, N, 0, E)).int()
    ia = torch.from_numpy(prng.randint(14, N, 0, E)).int()
    ia[ic == 1] = 0                                    # empty pair buckets (1, a > 0)
    off = torch.full((2 * E + 1,), -1, dtype=torch.int32, device="cuda")
    poff = torch.full((E * E + 1,), -1, dtype=torch.int32, device="cuda")
    perm = torch.full((2 * N + lib.vb_route_bucket_scratch_ints(N, E),), -1, dtype=torch.int32, device="cuda")
    ppa = torch.full((N,), -1, dtype=torch.int32, device="cuda")
    L.check(lib.vb_route_bucket_pairs(L.ptr(dev(ic)), L.ptr(dev(ia)), N, E, L.ptr(off), L.ptr(perm), L.ptr(poff), L.ptr(ppa), L.stream_ptr()), "bucket")
    sync()
    off, poff, perm, ppa = off.cpu().long(), poff.cpu().long(), perm.cpu()[:2 * N].long(), ppa.cpu().long()
    cnt = torch.cat([torch.bincount(ic.long(), minlength=E), torch.bincount(ia.long(), minlength=E)])
    assert torch.equal(off, torch.cat([torch.zeros(1, dtype=torch.long), cnt.cumsum(0)]))
    pair = ic.long() * E + ia.long()
    assert torch.equal(poff, torch.cat([torch.zeros(1, dtype=torch.long), torch.bincount(pair, minlength=E * E).cumsum(0)]))
    for g in range(E * E):
        assert torch.equal(perm[poff[g]:poff[g + 1]], (pair == g).nonzero().squeeze(1)), f"pair bucket {g}"
    for g in range(2 * E):                             # every expert group holds exactly its tokens (any order)
        src = ic if g < E else ia
        assert torch.equal(perm[off[g]:off[g + 1]].sort().values, (src == (g % E)).nonzero().squeeze(1)), f"group {g}"
    assert torch.equal(perm[ppa], perm[:N]) and int(ppa.min()) >= N and ppa.unique().numel() == N
    # acoustic half: acoustic-major order of the same pair buckets
    a_of_slot = ia.long()[perm[N:]]
    assert bool((a_of_slot[1:] >= a_of_slot[:-1]).all())


@pytest.mark.parametrize("npl", [1, 2])
def test_grouped_swiglu(lib, npl):
    N, D, H, G = 700, 768, 512, 4
    u = rnd((N, D), "su")
    w1, w3, w2 = rnd((G, H, D), "sw1", 0.04), rnd((G, H, D), "sw3", 0.04), rnd((G, D, H), "sw2", 0.04)
    idx = torch.from_numpy(prng.randint(5, N, 0, G)).int()
    scale = rnd((N,), "ss").abs() + 0.1
    off = torch.zeros(2 * G + 1, dtype=torch.int32, device="cuda")
    perm = torch.zeros(2 * N + lib.vb_route_bucket_scratch_ints(N, G), dtype=torch.int32, device="cuda")
    L.check(lib.vb_route_bucket(L.ptr(dev(idx)), L.ptr(dev(idx)), N, G, L.ptr(off), L.ptr(perm), L.stream_ptr()), "bucket")
    w13 = torch.stack([w1, w3], dim=2).reshape(G, 2 * H, D)
    up, w13p, w2p = dev(pack.to_planes(u, npl)), dev(pack.to_planes(w13, npl)), dev(pack.to_planes(w2, npl))
    hidden = torch.zeros(npl, N, H, dtype=torch.bfloat16, device="cuda")
    out = torch.full((N, D), float("nan"), device="cuda")
    L.check(lib.vb_grouped_swiglu(L.ptr(up), L.ptr(perm), L.ptr(off), G, N, L.ptr(w13p), L.ptr(w2p), L.ptr(dev(scale)), D, H, npl,
                                  L.ptr(hidden), L.ptr(out), L.stream_ptr()), "grouped_swiglu")
    sync()
    ref = torch.zeros(N, D, dtype=torch.float64)
    for g in range(G):
        sel = (idx == g).nonzero().squeeze(1)
        ref[sel] = ref_cpu.swiglu(u[sel].double(), w1[g].double(), w2[g].double(), w3[g].double()) * scale[sel].double().unsqueeze(1)
    tol = 8e-3 if npl == 1 else 4e-5
    assert rel_l2(out, ref) < tol, describe(f"grouped_swiglu np={npl}", out, ref)


def test_fill_gumbel_statistics_and_keying(lib):
    B, nb, T, W = 2, 2, 500, 4
    n = nb * B * T * W
    a = torch.zeros(n, device="cuda")
    b = torch.zeros(nb * 1 * T * W, device="cuda")
    L.check(lib.vb_fill_gumbel(L.ptr(a), B, nb, T, W, 99, 0, 3, 1, 2, L.stream_ptr()), "fill_gumbel")
    L.check(lib.vb_fill_gumbel(L.ptr(b), 1, nb, T, W, 99, 1, 3, 1, 2, L.stream_ptr()), "fill_gumbel")   # clip 1 alone
    sync()
    a = a.cpu().view(nb, B, T, W)
    assert abs(float(a.mean()) - 0.5772) < 0.05 and abs(float(a.var()) - 1.6449) < 0.15
    assert torch.isfinite(a).all()
    # draws are keyed by the GLOBAL clip index: clip 1 gets the same numbers wherever it sits in a batch
    assert torch.equal(a[:, 1], b.cpu().view(nb, 1, T, W)[:, 0])
    assert not torch.equal(a[:, 0], a[:, 1])


def test_cast_planes_roundtrip(lib):
    x = rnd((1000, 33), "cp", 3.0)
    out = torch.zeros(2, 1000, 33, dtype=torch.bfloat16, device="cuda")
    L.check(lib.vb_cast_planes(L.ptr(dev(x)), x.numel(), L.ptr(out), 2, L.stream_ptr()), "cast")
    sync()
    assert torch.equal(out.cpu(), pack.to_planes(x, 2))       # bit-exact with the host packer (RNE both sides)
    assert rel_l2(pack.planes_to_float(out.cpu()), x) < 2e-5


def test_staged_conv_epilogues_are_bit_identical(lib, monkeypatch):
    """The split-bf16 conv kernel and the fused ResBlock-pair kernel pass their accumulator tiles through wave-private LDS patches
    to move residual / output as 16-byte lane accesses when the rows are 16-B aligned (T % 4 == 0); VB_CONV_DIRECT_EPI=1 keeps the
    4-byte direct epilogue.  Same arithmetic per element: the whole vocoder must come out bit for bit the same (also on a
    length that is not a multiple of 4, where both take the direct path for the plain convs)."""
    from versband_amd import synth
    from versband_amd.engine import Context, build_hifigan
    hcfg = synth.HifiGanConfig()
    sdh = synth.make_state_dict(synth.hifigan_shapes(hcfg), 77)
    net = build_hifigan(Context("cuda:0"), sdh, hcfg.as_hparams())
    for T in (48, 37):
        mel = torch.from_numpy(prng.uniform(prng.key_seed(9, f"mel{T}"), 2 * 80 * T, -5.0, 1.0).reshape(2, 80, T)).cuda()
        a = net.run(mel).clone()
        monkeypatch.setenv("VB_CONV_DIRECT_EPI", "1")
        lib.vb_tune_reload()
        b = net.run(mel).clone()
        monkeypatch.delenv("VB_CONV_DIRECT_EPI")
        lib.vb_tune_reload()
        sync()
        assert torch.isfinite(a).all() and torch.equal(a, b), f"T={T}: staged vs direct epilogue differ by {float((a - b).abs().max()):.3e}"


# ---------------------------------------------------------------- exact-fp32 DMA-fed conv / fused pair (round 4) ------
G_CASES = [  # B, Ci, T, Co, k, dil, in_act, res  - every case takes conv1d_f32g_kernel (Ci % 16 == 0, Co % 4 == 0, T % 4 == 0)
    (2, 32, 1000, 32, 11, 5, 1, True),      # 32 x 256 tiles, ragged last tile, both clip ends inside a window
    (1, 64, 520, 64, 7, 3, 1, True),        # 64 x 128 tiles
    (2, 128, 388, 128, 3, 1, 1, True),      # 128 x 128 tiles, one channel tile
    (1, 256, 260, 256, 11, 1, 1, False),    # two channel tiles
    (2, 1536, 100, 768, 3, 1, 0, False),    # 96 chunks: the window / weight rings wrap many times
    (1, 384, 1504, 80, 5, 1, 0, False),     # Co < channel tile
    (1, 48, 132, 100, 1, 1, 0, True),       # k = 1: a new window every ring step; Co not a multiple of 32
    (1, 16, 64, 36, 2, 1, 0, False),        # one chunk, two taps
    (3, 128, 752, 1536, 3, 1, 0, False),    # 96-sample tiles win the tile choice at this shape
    (1, 32, 752, 1536, 3, 1, 0, True),      # one clip: 64 x 128 tiles (the launch would make 96 workgroups of 128 x 96)
    (1, 64, 752, 768, 3, 1, 1, True),       # one clip, 768 channels: 64 x 64 tiles
    (2, 1536, 752, 752, 1, 1, 0, False),    # the VAE's attention scores at two clips: 64 x 128 tiles on the 4-stage ring, a new window every step
    (1, 32, 752, 64, 1, 1, 1, True),        # k = 1, two chunks, Co <= 64
    (2, 128, 520, 128, 7, 3, 1, True),      # 128 x 128 tiles, seven taps unrolled (the ResBlock tap counts 3 / 7 / 11 are template parameters)
    (1, 256, 388, 256, 11, 5, 1, True),     # eleven taps, dilation 5, 16 chunks
    (3, 128, 752, 1536, 7, 1, 1, False),    # 96-sample tiles, seven taps
]


def _conv_f32(lib, x, wpk, b, r, B, Ci, T, Co, k, dil, pad, act, tr=(1, 0, 0), T_out=None):
    T_out = T if T_out is None else T_out
    out = torch.full((B, Co, T_out), float("nan"), device="cuda")
    L.check(lib.vb_conv1d_f32(L.ptr(x), L.ptr(wpk), L.ptr(b), B, Ci, T, Co, k, dil, pad, tr[0], tr[1], tr[2], T_out, act, 0.1,
                              L.ptr(r) if r is not None else None, L.ptr(out), None, 0, L.stream_ptr()), "conv")
    sync()
    return out


@pytest.mark.parametrize("B,Ci,T,Co,k,dil,act,res", G_CASES)
def test_fp32_dma_conv_is_bit_identical_to_register_staged(lib, monkeypatch, B, Ci, T, Co, k, dil, act, res):
    """conv1d_f32g_kernel (window and weight tiles by global_load_lds, LeakyReLU on the fragment, staged epilogue) keeps the chunk ->
    tap -> channel-pair order of conv1d_f32_kernel (VB_CONV_F32_OLD=1): same bits; and both are fp32-roundoff close to float64."""
    x, w, b = dev(rnd((B, Ci, T), "gx")), rnd((Co, Ci, k), "gw", 1.0 / (Ci * k) ** 0.5), dev(rnd((Co,), "gb"))
    r = dev(rnd((B, Co, T), "gr")) if res else None
    pad = (k - 1) * dil // 2
    wpk = dev(pack.pack_conv(w))
    T_out = T + 2 * pad - dil * (k - 1)
    new = _conv_f32(lib, x, wpk, b, r, B, Ci, T, Co, k, dil, pad, act, T_out=T_out) if T_out == T else None
    if new is None:      # even k: the output is one sample shorter; the residual must match it
        r = r[:, :, :T_out].contiguous() if r is not None else None
        new = _conv_f32(lib, x, wpk, b, r, B, Ci, T, Co, k, dil, pad, act, T_out=T_out)
    monkeypatch.setenv("VB_CONV_F32_OLD", "1")
    lib.vb_tune_reload()
    old = _conv_f32(lib, x, wpk, b, r, B, Ci, T, Co, k, dil, pad, act, T_out=T_out)
    monkeypatch.delenv("VB_CONV_F32_OLD")
    lib.vb_tune_reload()
    assert torch.isfinite(new).all() and torch.equal(new, old), f"DMA-fed vs register-staged differ by {float((new - old).abs().max()):.3e}"
    xin = F.leaky_relu(x.double().cpu(), 0.1) if act else x.double().cpu()
    ref = F.conv1d(xin, w.double(), b.double().cpu(), dilation=dil, padding=pad)
    if res:
        ref = ref + r.double().cpu()
    assert rel_l2(new, ref) < 2e-6, describe("conv1d fp32 dma", new, ref)


MF_CASES = [(2, 128, 1000, 128, 3, 1, 1, True), (1, 256, 752, 256, 3, 3, 1, False), (2, 128, 488, 128, 3, 5, 1, True),
            (1, 128, 1204, 128, 7, 1, 1, True), (2, 256, 360, 256, 7, 3, 1, False), (1, 128, 600, 128, 7, 5, 0, True),
            (1, 256, 724, 256, 11, 1, 1, True), (2, 128, 500, 128, 11, 3, 1, False), (1, 128, 1000, 128, 11, 5, 1, True),
            (1, 64, 2000, 128, 11, 5, 1, True), (1, 64, 2000, 64, 11, 5, 1, True), (2, 64, 1028, 64, 7, 3, 1, True), (1, 64, 520, 64, 3, 1, 1, False), (1, 384, 304, 192, 3, 1, 0, True), (2, 96, 244, 160, 5, 2, 1, False), (1, 128, 120, 128, 7, 3, 1, True),
            (1, 384, 1504, 80, 5, 1, 0, False), (1, 128, 8, 128, 3, 1, 1, True), (2, 128, 60, 128, 11, 5, 1, True), (1, 16, 400, 64, 3, 1, 1, False),
            (3, 64, 132, 192, 7, 1, 0, True)]


@pytest.mark.parametrize("B,Ci,T,Co,k,dil,act,res", MF_CASES)
def test_fp32_minimal_filtering_conv_matches_the_direct_kernel(lib, B, Ci, T, Co, k, dil, act, res):
    """conv1d_f32w_kernel (round 6): the HiFi-GAN ResBlock convolutions (vocoder/hifigan/modules/hifigan.py:27-64) as F(2,3) minimal
    filtering - pseudo-tap weights pre-combined by pack.pack_conv_mf, outputs formed in the epilogue.  fp32 products on the f32 MFMA,
    1.4-1.5x fewer of them; NOT bit-identical to the direct kernel but equally close to float64: both within 2e-6 (rel-L2), and
    the two differ by < 4e-6 of the output's max (measured 0.5-2e-6: two fp32 summation orders).  Covers every (k, dilation) of the generator, partial last tiles, T not a multiple of the
    tile, clip ends (padding), Ci != Co, the accumulate-into form and k = 5 (group + pair remainder)."""
    x, w, b = dev(rnd((B, Ci, T), "mx")), rnd((Co, Ci, k), "mw", 1.0 / (Ci * k) ** 0.5), dev(rnd((Co,), "mb"))
    r = dev(rnd((B, Co, T), "mr")) if res else None
    pad = (k - 1) * dil // 2
    wpk, wmf = dev(pack.pack_conv(w)), dev(pack.pack_conv_mf(w))
    assert wmf.shape == (pack.mf_pseudo_taps(k), Ci, Co)
    direct = _conv_f32(lib, x, wpk, b, r, B, Ci, T, Co, k, dil, pad, act)
    old = dev(rnd((B, Co, T), "mo"))
    for alpha, beta in ((1.0, 0.0), (1.0 / 3, 1.0)):
        out = old.clone() if beta else torch.full((B, Co, T), float("nan"), device="cuda")
        L.check(lib.vb_conv1d_f32_mf(L.ptr(x), L.ptr(wpk), L.ptr(wmf), L.ptr(b), B, Ci, T, Co, k, dil, pad, T, act, 0.1,
                                     L.ptr(r) if r is not None else None, alpha, beta, L.ptr(out), L.stream_ptr()), "conv mf")
        sync()
        xin = F.leaky_relu(x.double().cpu(), 0.1) if act else x.double().cpu()
        ref = F.conv1d(xin, w.double(), b.double().cpu(), dilation=dil, padding=pad)
        if res:
            ref = ref + r.double().cpu()
        full = ref
        ref = alpha * ref + beta * old.double().cpu() if beta else ref
        assert torch.isfinite(out).all()
        assert rel_l2(out, ref) < 2e-6, describe(f"conv1d fp32 mf alpha={alpha:.3f}", out, ref)
        if not beta:
            assert rel_l2(direct, full) < 2e-6
            assert float((out - direct).abs().max()) < 4e-6 * float(full.abs().max()), describe("mf vs direct", out, direct)


def test_fp32_minimal_filtering_switches(lib, monkeypatch):
    """VB_MF_OCC=3 runs the same kernel built for three workgroups per CU: same bits as the default two-per-CU build.  VB_CONV_MF_OFF=1 ignores
    the minimal-filtering weights: the call is the direct fp32 kernel, bit for bit."""
    B, C, T, k, dil = 2, 128, 1000, 11, 3
    x, w, b = dev(rnd((B, C, T), "sx2")), rnd((C, C, k), "sw2", 1.0 / (C * k) ** 0.5), dev(rnd((C,), "sb2"))
    pad = (k - 1) * dil // 2
    wpk, wmf = dev(pack.pack_conv(w)), dev(pack.pack_conv_mf(w))

    def mf():
        out = torch.full((B, C, T), float("nan"), device="cuda")
        L.check(lib.vb_conv1d_f32_mf(L.ptr(x), L.ptr(wpk), L.ptr(wmf), L.ptr(b), B, C, T, C, k, dil, pad, T, 1, 0.1, None, 1.0, 0.0,
                                     L.ptr(out), L.stream_ptr()), "conv mf")
        sync()
        return out

    direct = _conv_f32(lib, x, wpk, b, None, B, C, T, C, k, dil, pad, 1)
    base = mf()
    assert not torch.equal(base, direct) and float((base - direct).abs().max()) < 4e-6 * float(direct.abs().max())
    try:
        monkeypatch.setenv("VB_MF_OCC", "3")
        lib.vb_tune_reload()
        assert torch.equal(mf(), base)
        monkeypatch.delenv("VB_MF_OCC")
        monkeypatch.setenv("VB_CONV_MF_OFF", "1")
        lib.vb_tune_reload()
        assert torch.equal(mf(), direct)
    finally:
        monkeypatch.delenv("VB_MF_OCC", raising=False)
        monkeypatch.delenv("VB_CONV_MF_OFF", raising=False)
        lib.vb_tune_reload()


def test_one_tap_conv_is_stable_beside_a_second_gpu_process(lib):
    """Round 5: in front of the second-last ring step of a 1-tap layer on the 4-stage weight ring the counted vmcnt wait let the window's
    last DMA piece fly (conv1d_f32g.hip, `lag`): whole wrong 64 x 128 tiles, but only while something else kept the memory system busy -
    10-13 of 30 runs beside a second process, none alone (profiles/r05_conv_tail_race.txt), which is how two ranks sharing a GPU
    found it.  The shape of the VAE's attention scores at two clips (1536 -> 752 channels, k = 1: the tile choice takes 64 x 128 on the
    4-stage ring), repeated beside a process that streams HBM: every run has the first run's bits."""
    from tests.helpers import beside_load
    B, Ci, T, Co = 2, 1536, 752, 752
    x, w, b = dev(rnd((B, Ci, T), "sx")), rnd((Co, Ci, 1), "sw", 1.0 / Ci ** 0.5), dev(rnd((Co,), "sb"))
    wpk = dev(pack.pack_conv(w))
    with beside_load(45) as load:
        first = _conv_f32(lib, x, wpk, b, None, B, Ci, T, Co, 1, 1, 0, 0)
        ref = F.conv1d(x.double().cpu(), w.double(), b.double().cpu())
        assert rel_l2(first, ref) < 2e-6, describe("1x1 conv fp32", first, ref)
        bad = [i for i in range(200) if not torch.equal(_conv_f32(lib, x, wpk, b, None, B, Ci, T, Co, 1, 1, 0, 0), first)]
        assert not bad, f"runs {bad} of 200 differ from the first"
        assert load.alive(), "the load process ended before the repeats did"


@pytest.mark.parametrize("B,Ci,T,Co,k,u", [(2, 512, 24, 256, 16, 8), (1, 256, 32, 128, 15, 5), (2, 64, 52, 32, 4, 2), (1, 128, 20, 64, 8, 4)])
def test_fp32_dma_conv_transpose_is_bit_identical(lib, monkeypatch, B, Ci, T, Co, k, u):
    x, w, b = dev(rnd((B, Ci, T), "hx")), rnd((Ci, Co, k), "hw", (u / (Ci * k)) ** 0.5), dev(rnd((Co,), "hb"))
    p = (k - u) // 2
    ref = F.conv_transpose1d(F.leaky_relu(x.double().cpu(), 0.1), w.double(), b.double().cpu(), stride=u, padding=p)
    T_out = ref.shape[-1]
    wpk = dev(pack.pack_conv_transpose(w, u))
    new = _conv_f32(lib, x, wpk, b, None, B, Ci, T, Co, 0, 1, 0, 1, tr=(u, p, k), T_out=T_out)
    monkeypatch.setenv("VB_CONV_F32_OLD", "1")
    lib.vb_tune_reload()
    old = _conv_f32(lib, x, wpk, b, None, B, Ci, T, Co, 0, 1, 0, 1, tr=(u, p, k), T_out=T_out)
    monkeypatch.delenv("VB_CONV_F32_OLD")
    lib.vb_tune_reload()
    assert torch.isfinite(new).all() and torch.equal(new, old)
    assert rel_l2(new, ref) < 2e-6, describe("conv_transpose1d fp32 dma", new, ref)


@pytest.mark.parametrize("B,C,T,k,dil,alpha,beta", [(2, 32, 1000, 3, 1, 1.0, 0.0), (1, 32, 472, 11, 5, 1.0 / 3, 1.0), (2, 64, 600, 7, 3, 1.0, 0.0),
                                                   (1, 64, 244, 11, 5, 1.0 / 3, 1.0), (1, 64, 128, 3, 5, 1.0 / 3, 0.0),
                                                   (1, 128, 360, 7, 1, 1.0, 0.0), (2, 128, 120, 11, 3, 1.0 / 3, 1.0),
                                                   (2, 32, 1000, 7, 3, 1.0, 0.0), (1, 32, 360, 5, 1, 1.0, 0.0), (2, 64, 372, 3, 1, 1.0, 0.0),
                                                   (1, 64, 500, 5, 3, 1.0, 0.0)])
def test_fp32_respair_equals_two_fp32_convolutions(lib, B, C, T, k, dil, alpha, beta):
    """respair_f32_kernel (vocoder/hifigan/modules/hifigan.py:27-64 ResBlock1 pair, intermediate kept in LDS) against the two launches
    of the fp32 convolution kernel it replaces: same accumulation order, same epilogue arithmetic - equal bit for bit (torch.equal
    treats the two zeros alike), incl. the accumulate-into-the-MRF-sum form (alpha = 1/3, beta = 1) and both clip ends."""
    x = dev(rnd((B, C, T), "px"))
    w1, w2 = rnd((C, C, k), "pw1", 1.0 / (C * k) ** 0.5), rnd((C, C, k), "pw2", 1.0 / (C * k) ** 0.5)
    b1, b2 = dev(rnd((C,), "pb1")), dev(rnd((C,), "pb2"))
    acc0 = dev(rnd((B, C, T), "pacc"))
    p1, p2 = dev(pack.pack_conv(w1)), dev(pack.pack_conv(w2))
    fused = acc0.clone()
    L.check(lib.vb_respair_f32(L.ptr(x), L.ptr(p1), L.ptr(b1), L.ptr(p2), L.ptr(b2), B, C, T, k, dil, 0.1, alpha, beta, L.ptr(fused),
                               L.stream_ptr()), "respair_f32")
    sync()
    t1 = _conv_f32(lib, x, p1, b1, None, B, C, T, C, k, dil, (k - 1) * dil // 2, 1)
    t2 = _conv_f32(lib, t1, p2, b2, x, B, C, T, C, k, 1, (k - 1) // 2, 1)
    ref64 = F.conv1d(F.leaky_relu(F.conv1d(F.leaky_relu(x.double().cpu(), 0.1), w1.double(), b1.double().cpu(), dilation=dil,
                                           padding=(k - 1) * dil // 2), 0.1), w2.double(), b2.double().cpu(), padding=(k - 1) // 2) + x.double().cpu()
    ref64 = alpha * ref64 + beta * acc0.double().cpu()
    assert torch.isfinite(fused).all()
    assert rel_l2(fused, ref64) < 2e-6, describe("respair fp32", fused, ref64)
    if alpha == 1.0 and beta == 0.0:
        assert torch.equal(fused, t2), f"fused pair vs two launches differ by {float((fused - t2).abs().max()):.3e}"


@pytest.mark.parametrize("C", [32, 64])
@pytest.mark.parametrize("B,T,k,dil,alpha,beta", [(2, 1000, 3, 1, 1.0, 0.0), (1, 472, 11, 5, 1.0 / 3, 1.0), (2, 1000, 7, 3, 1.0, 0.0), (1, 2048, 11, 1, 1.0, 0.0),
                                                 (1, 228, 11, 5, 1.0, 0.0), (2, 252, 3, 5, 1.0, 0.0), (1, 60, 7, 1, 1.0 / 3, 1.0), (1, 1504, 11, 3, 1.0, 0.0),
                                                 (1, 736, 3, 3, 1.0, 0.0), (3, 244, 7, 5, 1.0, 0.0), (1, 120, 11, 5, 1.0, 0.0), (1, 116, 11, 1, 1.0, 0.0)])
def test_fp32_minimal_filtering_pair_matches_float64_and_the_direct_pair(lib, C, B, T, k, dil, alpha, beta):
    """respair_f32w_kernel (round 6: the 32-channel ResBlock1 pair with F(2,3) minimal filtering in both convolutions, intermediate in LDS)
    against float64 to fp32 roundoff and against the direct fused pair (< 4e-6 of the output's max); workgroup runs of 256 / 240 intermediate
    positions (d = 1 / 3, 5), both clip ends, T smaller than one run, the accumulate-into-the-MRF-sum form.  It is NOT bit-identical to two
    conv1d_f32w launches: F(2,3) forms the even and the odd output of a pair by different sums, and which member a position is depends on where
    the pairs start - the unfused kernel's start at multiples of its tile, the fused kernel's intermediate run starts (k - 1) / 2 positions in
    front of its outputs - so ~1/3 of the elements differ by one ulp (measured); both are the same distance from float64."""
    x = dev(rnd((B, C, T), "wx"))
    w1, w2 = rnd((C, C, k), "ww1", 1.0 / (C * k) ** 0.5), rnd((C, C, k), "ww2", 1.0 / (C * k) ** 0.5)
    b1, b2 = dev(rnd((C,), "wb1")), dev(rnd((C,), "wb2"))
    acc0 = dev(rnd((B, C, T), "wacc"))
    p1, p2, m1, m2 = dev(pack.pack_conv(w1)), dev(pack.pack_conv(w2)), dev(pack.pack_conv_mf(w1)), dev(pack.pack_conv_mf(w2))
    fused = acc0.clone()
    L.check(lib.vb_respair_f32_mf(L.ptr(x), L.ptr(m1), L.ptr(b1), L.ptr(m2), L.ptr(b2), B, C, T, k, dil, 0.1, alpha, beta, L.ptr(fused),
                                  L.stream_ptr()), "respair_f32_mf")
    sync()
    ref64 = F.conv1d(F.leaky_relu(F.conv1d(F.leaky_relu(x.double().cpu(), 0.1), w1.double(), b1.double().cpu(), dilation=dil,
                                           padding=(k - 1) * dil // 2), 0.1), w2.double(), b2.double().cpu(), padding=(k - 1) // 2) + x.double().cpu()
    ref64 = alpha * ref64 + beta * acc0.double().cpu()
    assert torch.isfinite(fused).all()
    assert rel_l2(fused, ref64) < 2e-6, describe("respair fp32 mf", fused, ref64)
    if alpha == 1.0 and beta == 0.0:
        def mf(xin, wpk, wmf, bias, d, act, res):
            out = torch.full((B, C, T), float("nan"), device="cuda")
            L.check(lib.vb_conv1d_f32_mf(L.ptr(xin), L.ptr(wpk), L.ptr(wmf), L.ptr(bias), B, C, T, C, k, d, (k - 1) * d // 2, T, act, 0.1,
                                         L.ptr(res) if res is not None else None, 1.0, 0.0, L.ptr(out), L.stream_ptr()), "conv mf")
            sync()
            return out
        t1 = F.leaky_relu(mf(x, p1, m1, b1, dil, 1, None), 0.1)
        t2 = mf(t1, p2, m2, b2, 1, 0, x)
        assert float((fused - t2).abs().max()) < 4e-6 * float(ref64.abs().max()) and rel_l2(t2, ref64) < 2e-6
        direct = acc0.clone()
        L.check(lib.vb_respair_f32(L.ptr(x), L.ptr(p1), L.ptr(b1), L.ptr(p2), L.ptr(b2), B, C, T, k, dil, 0.1, alpha, beta, L.ptr(direct),
                                   L.stream_ptr()), "respair_f32")
        sync()
        assert float((fused - direct).abs().max()) < 4e-6 * float(ref64.abs().max())


@pytest.mark.parametrize("C,k,dil", [(32, 7, 3), (32, 11, 5), (64, 3, 1), (64, 7, 3), (64, 11, 5)])
def test_fp32_respair_unrolled_control_flow_equals_the_runtime_loop(lib, monkeypatch, C, k, dil):
    """respair_f32_kernel with the kernel size as a template parameter (a chunk's steps unrolled, wait counts / tap offsets / the tile to
    request as immediates) against its runtime loop (VB_CONV_F32_RT_TAPS=1): same tiles in the same order - the same bits, incl. both clip
    ends and the accumulate form."""
    B, T = 2, 1100
    x = dev(rnd((B, C, T), "qx"))
    p1, p2 = dev(pack.pack_conv(rnd((C, C, k), "qw1", 1.0 / (C * k) ** 0.5))), dev(pack.pack_conv(rnd((C, C, k), "qw2", 1.0 / (C * k) ** 0.5)))
    b1, b2 = dev(rnd((C,), "qb1")), dev(rnd((C,), "qb2"))
    acc0 = dev(rnd((B, C, T), "qacc"))

    def run():
        out = acc0.clone()
        L.check(lib.vb_respair_f32(L.ptr(x), L.ptr(p1), L.ptr(b1), L.ptr(p2), L.ptr(b2), B, C, T, k, dil, 0.1, 1.0 / 3, 1.0, L.ptr(out),
                                   L.stream_ptr()), "respair_f32")
        sync()
        return out
    new = run()
    monkeypatch.setenv("VB_CONV_F32_RT_TAPS", "1")
    lib.vb_tune_reload()
    old = run()
    monkeypatch.delenv("VB_CONV_F32_RT_TAPS")
    lib.vb_tune_reload()
    assert torch.isfinite(new).all() and torch.equal(new, old), f"unrolled vs runtime loop differ by {float((new - old).abs().max()):.3e}"


def test_fp32_vocoder_fused_pairs_equal_unfused_launches(lib, monkeypatch):
    """The whole generator in exact fp32 (BASELINE configs[1]: "fp32 vocoder"): ResBlock1 pairs fused at 32 / 64 channels
    (respair_f32_kernel, incl. the accumulate-into-the-MRF-sum pairs with alpha = 1/3, beta = 1) against one launch per convolution
    (VB_FP32_PAIRS=""), and the DMA-fed kernel against the register-staged one (VB_CONV_F32_OLD=1): the waveform is the same bit for bit."""
    from versband_amd import synth
    from versband_amd.engine import Context, build_hifigan
    hcfg = synth.HifiGanConfig()
    sdh = synth.make_state_dict(synth.hifigan_shapes(hcfg), 78)
    ctx = Context("cuda:0")
    mel = torch.from_numpy(prng.uniform(prng.key_seed(9, "melf"), 2 * 80 * 44, -5.0, 1.0).reshape(2, 80, 44)).cuda()
    fused = build_hifigan(ctx, sdh, hcfg.as_hparams(), precision="fp32").run(mel).clone()
    monkeypatch.setenv("VB_FP32_PAIRS", "")
    unfused_net = build_hifigan(ctx, sdh, hcfg.as_hparams(), precision="fp32")
    monkeypatch.delenv("VB_FP32_PAIRS")
    unfused = unfused_net.run(mel).clone()
    monkeypatch.setenv("VB_CONV_F32_OLD", "1")
    lib.vb_tune_reload()
    old = unfused_net.run(mel).clone()
    monkeypatch.delenv("VB_CONV_F32_OLD")
    lib.vb_tune_reload()
    sync()
    assert torch.isfinite(fused).all()
    assert torch.equal(unfused, old), f"DMA-fed vs register-staged vocoder differ by {float((unfused - old).abs().max()):.3e}"
    assert torch.equal(fused, unfused), f"fused pairs vs unfused differ by {float((fused - unfused).abs().max()):.3e}"


@pytest.mark.parametrize("B,Ci,T,k,dil,act", [(2, 32, 1000, 7, 1, 1), (1, 32, 1541, 7, 1, 1), (1, 80, 300, 5, 3, 0), (3, 16, 64, 1, 1, 0)])
def test_one_output_channel_conv_is_bit_identical_to_the_mfma_kernel(lib, monkeypatch, B, Ci, T, k, dil, act):
    """conv1d_co1_kernel (HiFi-GAN conv_post: one output channel) runs the fmaf chain of v_mfma_f32_32x32x2_f32 on the vector ALU in the
    same chunk -> tap -> channel order: same bits as the MFMA kernel it replaces (VB_CONV_F32_OLD=1), incl. lengths that are not a
    multiple of anything and a channel count that is not a multiple of the 16-channel chunk."""
    x, w, b = dev(rnd((B, Ci, T), "ox")), rnd((1, Ci, k), "ow", 1.0 / (Ci * k) ** 0.5), dev(rnd((1,), "ob"))
    pad = (k - 1) * dil // 2
    wpk = dev(pack.pack_conv(w))
    new = _conv_f32(lib, x, wpk, b, None, B, Ci, T, 1, k, dil, pad, act)
    monkeypatch.setenv("VB_CONV_F32_OLD", "1")
    lib.vb_tune_reload()
    old = _conv_f32(lib, x, wpk, b, None, B, Ci, T, 1, k, dil, pad, act)
    monkeypatch.delenv("VB_CONV_F32_OLD")
    lib.vb_tune_reload()
    assert torch.isfinite(new).all() and torch.equal(new, old), f"VALU vs MFMA one-channel conv differ by {float((new - old).abs().max()):.3e}"
    xin = F.leaky_relu(x.double().cpu(), 0.1) if act else x.double().cpu()
    ref = F.conv1d(xin, w.double(), b.double().cpu(), dilation=dil, padding=pad)
    assert rel_l2(new, ref) < 2e-6
