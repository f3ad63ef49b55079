"""Tensor-level wrappers over the C-ABI (include/tem_hip.h).

Every function takes torch CUDA tensors only to obtain device pointers, strides and the
current HIP stream -- all arithmetic happens in libtem_hip.so.  Activations are 5-D
channels-last views `t[N, D, H, W, C]` with `t.stride(4) == 1`; `t.stride(3)` (the leading
dimension `ld`) may exceed C so that a tensor can be a channel slice of a concat buffer.
There is no CPU fallback: CPU tensors raise.
"""
import ctypes
from typing import Optional, Sequence, Tuple

import torch

from . import _lib

ACT = {None: 0, "none": 0, "relu": 1, "sigmoid": 2}

# Optional live kernel timing (bench.py): a list that receives (tag, flops, start_event, end_event)
# for every convolution launch; events are recorded on the stream the kernel is launched on.
PROFILER = None
# When set, only launches whose kernel tag is in this set are timed: two HIP events per launch are not free (a
# fully instrumented cfg-2 step has ~200 of them and runs 3.5-6 ms slower), so bench.py instruments everything in a
# warm-up step and only the dominant kernel inside the timed region.
PROFILER_FILTER = None


def _fwd_tag(mfma, k, cout, pp=False):
    if pp in (3, 4):  # the z-reuse team kernel (csrc/conv_zr.hip); 4 = its split-K launch (16^3 / 32^3 levels)
        return f"k_conv_zr_{'f16x3' if int(mfma) == 4 else 'f16' if int(mfma) == 5 else 'bf16' if int(mfma) == 7 else 'fp32' if int(mfma) == 1 else 'bf16x3'}<3,3,3>"
    if pp:  # the ping-pong team kernel (csrc/conv_pp.hip)
        return f"k_conv_pp_{'f16x3' if int(mfma) == 4 else 'bf16x3'}<{k[0]},{k[1]},{k[2]},CT={2 if cout % 64 == 0 else 1}>"
    kind = ({2: "k_conv_fwd_bf16x3", 3: "k_conv_fwd_bf16x6", 4: "k_conv_fwd_f16x3", 5: "k_conv_fwd_f16",
             6: "k_conv_fwd_f16x3", 7: "k_conv_fwd_bf16"}.get(int(mfma)) or
            ("k_conv_fwd_mfma" if mfma else "k_conv_fwd_valu")) + f"<{k[0]},{k[1]},{k[2]}"
    return kind + (f",NR={2 if cout % 64 == 0 else 1}>" if mfma else ">")


def _wgrad_tag(mfma, k, cout):
    ntaps = k[0] * k[1] * k[2]
    return ("k_conv_wgrad_bf16x3" if int(mfma) == 2 else "k_conv_wgrad_f16" if int(mfma) == 5 else
            "k_conv_wgrad_bf16" if int(mfma) == 7 else "k_conv_wgrad_f16x2" if int(mfma) == 8 else
            "k_conv_wgrad_mfma" if mfma else "k_conv_wgrad_valu") + \
        f"<{k[0]},{k[1]},{k[2]}" + (f",NCO={2 if cout >= 64 else 1}" if int(mfma) in (2, 5, 7, 8) and ntaps > 1 else "") + ">(+reduce)"


def _prof_begin(t, tag=None):
    if PROFILER is None or (PROFILER_FILTER is not None and tag not in PROFILER_FILTER):
        return None
    ev = torch.cuda.Event(enable_timing=True)
    ev.record(torch.cuda.current_stream(t.device))
    return ev


def _prof_end(t, ev0, tag, flops):
    if ev0 is None:
        return
    ev1 = torch.cuda.Event(enable_timing=True)
    ev1.record(torch.cuda.current_stream(t.device))
    PROFILER.append((tag, flops, ev0, ev1))


def _stream(t: torch.Tensor):
    """The stream the launch goes to: torch's current stream of the TENSOR's device.  A HIP launch is issued on the
    calling thread's current device, so that device is made current for the launch and `_lib.check()` -- which wraps
    every launch -- restores the caller's (one process drives one GPU in this design; a process that touches several --
    `DefaultTrainer(device="cuda:1")`, `predict_with_halo(gpu_ids=[1])` -- gets the device of the tensors it passes,
    like a torch op would, and keeps its own current device)."""
    _lib.launch_on(t.device.index)
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


# activation storage types of the C-ABI (TEM_ST_F32 / _F16 / _BF16, include/tem_hip.h): the element type of a tensor handed to
# the `_st` entry points, and -- shifted into the high bits of `use_mfma` -- to the convolution entry points
_ST = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def _st(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else _ST[t.dtype]


def _mode(mfma, x, y) -> int:
    """use_mfma | TEM_MFMA_STX(storage of x) | TEM_MFMA_STY(storage of y)"""
    return int(mfma) | (_st(x) << 8) | (_st(y) << 12)


def _same_st(*ts):
    sts = {t.dtype for t in ts if t is not None}
    if len(sts) > 1:
        raise ValueError(f"tensors of one call must share their storage type, got {sorted(str(d) for d in sts)}")


def _req_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "torch_em_amd ops run on MI355X only (got a CPU tensor); there is no CPU fallback for this path"
            )


def _act5(t: torch.Tensor) -> Tuple[int, int, int, int, int, int]:
    """(N, D, H, W, C, ld) of a channels-last 5-D view; validates the stride contract:
    element (n,z,y,x,c) at (((n*D+z)*H+y)*W+x)*ld + c."""
    if isinstance(t, Planar):
        return (*t.shape, 32)
    if isinstance(t, Probe):
        return (*t.shape, t.shape[4])
    if t.dim() != 5:
        raise ValueError(f"expected a 5-D NDHWC tensor, got shape {tuple(t.shape)}")
    if t.dtype not in _ST:
        raise ValueError(f"expected float32 (or float16 / bfloat16 activation storage), got {t.dtype}")
    N, D, H, W, C = t.shape
    s = t.stride()
    sizes = (N, D, H, W)
    inner = [1, 1, 1, 1]  # voxels spanned by one step along each dim
    for k in (2, 1, 0):
        inner[k] = inner[k + 1] * sizes[k + 1]
    ld = None
    for k in (3, 2, 1, 0):
        if sizes[k] > 1:
            if s[k] % inner[k]:
                raise ValueError(f"tensor is not a channels-last NDHWC view: shape {tuple(t.shape)}, strides {s}")
            ld = s[k] // inner[k]
            break
    if ld is None:
        ld = C
    ok = (C == 1 or s[4] == 1) and ld >= C
    for k in range(4):
        if sizes[k] > 1 and s[k] != ld * inner[k]:
            ok = False
    if not ok:
        raise ValueError(f"tensor is not a channels-last NDHWC view: shape {tuple(t.shape)}, strides {s}")
    return N, D, H, W, C, ld


def new_act(N, D, H, W, C, device, dtype=torch.float32) -> torch.Tensor:
    return torch.empty((N, D, H, W, C), dtype=dtype, device=device)


class Planar:
    """A [N, D, H, W, 2 * 32] activation whose two 32-channel halves are two DENSE tensors of one allocation
    (buf [2, N, D, H, W, 32]): the concat buffer of a level whose halves would otherwise be 64 bytes of every 128-byte line
    in 16-bit storage (DESIGN.md 6.R5 "half lines").  The kernels that read / write all 64 channels take it through a chunk
    stride (x_cs / y_cs of tem_conv3d_fwd_ex / _wgrad_ex); everything else gets `halves[i]`, an ordinary dense tensor."""

    def __init__(self, buf: torch.Tensor):
        if buf.dim() != 6 or buf.shape[0] != 2 or buf.shape[5] != 32 or not buf.is_contiguous() or buf.dtype not in (torch.float16, torch.bfloat16):
            raise ValueError("Planar: expects a contiguous 16-bit [2, N, D, H, W, 32] buffer")
        self.buf, self.halves = buf, (buf[0], buf[1])
        self.shape = (*buf.shape[1:5], 64)
        self.dtype, self.device, self.is_cuda = buf.dtype, buf.device, buf.is_cuda
        self.cs = buf[0].numel()      # elements between the two 32-channel chunks of a voxel

    @staticmethod
    def empty(N, D, H, W, device, dtype):
        return Planar(torch.empty((2, N, D, H, W, 32), dtype=dtype, device=device))

    def empty_like(self):
        return Planar(torch.empty_like(self.buf))

    def data_ptr(self):
        return self.buf.data_ptr()

    def dim(self):
        return 5

    def record_stream(self, s):
        self.buf.record_stream(s)


def _cs(t) -> int:
    return t.cs if isinstance(t, Planar) else 0


class Probe:
    """shape + storage type of a dense activation, for the query functions (conv_fwd_family, ..._ok) only"""

    def __init__(self, N, D, H, W, C, dtype):
        self.shape, self.dtype = (N, D, H, W, C), dtype


# ---------------------------------------------------------------- layout ----
def nchw_to_nhwc(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[N, C, *spatial] contiguous -> NDHWC (spatial may be 2-D or 3-D; 2-D gets D == 1)."""
    _req_cuda(x)
    x = x.contiguous()
    N, C = x.shape[:2]
    sp = tuple(x.shape[2:])
    D, H, W = (1,) * (3 - len(sp)) + sp
    if C == 1 and out is None:
        return x.reshape(N, D, H, W, 1)
    if out is None:
        out = new_act(N, D, H, W, C, x.device)
    _, _, _, _, _, ld = _act5(out)
    lib = _lib.load()
    _lib.check(lib.tem_nchw_to_nhwc(_p(x), _p(out), ld, N, C, D * H * W, _stream(x)), "tem_nchw_to_nhwc")
    return out


def nhwc_to_nchw(x: torch.Tensor) -> torch.Tensor:
    _req_cuda(x)
    N, D, H, W, C, ld = _act5(x)
    out = torch.empty((N, C, D, H, W), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    _lib.check(lib.tem_nhwc_to_nchw(_p(x), ld, _p(out), N, C, D * H * W, _stream(x)), "tem_nhwc_to_nchw")
    return out


# ------------------------------------------------------------------ conv ----
def mfma_ok(cin: int, cout: int, k: Sequence[int], wgrad: bool = False) -> bool:
    key = tuple(int(v) for v in k)
    if key not in ((3, 3, 3), (1, 3, 3), (1, 1, 1)):
        return False
    if wgrad:
        return cin % 32 == 0 and cout % 32 == 0
    return cin % 16 == 0 and cout % 32 == 0


def pack_weights(w: torch.Tensor, transpose: bool, mfma) -> torch.Tensor:
    """state_dict layout [Cout, Cin, (kd,) kh, kw] -> kernel layout (see tem_hip.h).
    mfma: False/0 generic, True/1 exact-fp32 MFMA fragments, 2 / 3 split-bf16 fragments (2 / 3 terms), 4 split-fp16
    (lo plane scaled), 5 one fp16 term (mixed precision), 6 split-fp16 with prescaled operands, 7 one bf16 term
    (mixed precision with dtype bfloat16)."""
    _req_cuda(w)
    w = w.detach().contiguous()
    cout, cin = w.shape[:2]
    k = tuple(w.shape[2:])
    k = (1,) * (3 - len(k)) + k
    lib = _lib.load()
    dst = torch.empty(lib.tem_conv_packed_size(cout, cin, k[0], k[1], k[2]), dtype=torch.float32, device=w.device)
    _lib.check(lib.tem_conv_pack_weights(_p(w), _p(dst), cout, cin, k[0], k[1], k[2], int(transpose),
                                         int(mfma), _stream(w)), "tem_conv_pack_weights")
    return dst


def pack_table(jobs):
    """Device descriptor tables for tem_conv_pack_weights_tiles (all tensors whose kernel has <= 27 taps: one workgroup per
    32 x 32 x taps tile) and tem_conv_pack_weights_batch (the rest).  jobs: (w, dst, cout, cin, k3, transpose, nsplit, fp16)."""
    import struct
    tiled, gather = b"", b""
    ntile = nitem = n_t = n_g = 0
    for w, dst, cout, cin, k, transpose, nsplit, fp16 in jobs:
        taps = k[0] * k[1] * k[2]
        if nsplit != 0 and taps <= 27 and cout % 16 == 0 and cin % 16 == 0:   # nsplit 0 = generic fp32 layout: gather kernel only
            tiled += struct.pack("<qq8iq", w.data_ptr(), dst.data_ptr(), cout, cin, k[0], k[1], k[2], int(transpose), nsplit,
                                 fp16, ntile)
            ntile += ((cout + 31) // 32) * ((cin + 31) // 32)
            n_t += 1
        else:
            gather += struct.pack("<qq8iq", w.data_ptr(), dst.data_ptr(), cout, cin, k[0], k[1], k[2], int(transpose),
                                  nsplit, fp16, nitem)
            nitem += cout * cin * taps // 8
            n_g += 1
    dev = jobs[0][0].device
    mk = lambda blob: torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev) if blob else None  # noqa: E731
    return {"tiles": mk(tiled), "n_tiles": n_t, "total_tiles": ntile, "table": mk(gather), "n": n_g, "total": nitem,
            "keep": [(j[0], j[1]) for j in jobs]}


def pack_weights_batch(tab):
    lib = _lib.load()
    if tab["n_tiles"]:
        _lib.check(lib.tem_conv_pack_weights_tiles(_p(tab["tiles"]), tab["n_tiles"], tab["total_tiles"],
                                                   _stream(tab["tiles"])), "tem_conv_pack_weights_tiles")
    if tab["n"]:
        _lib.check(lib.tem_conv_pack_weights_batch(_p(tab["table"]), tab["n"], tab["total"], _stream(tab["table"])),
                   "tem_conv_pack_weights_batch")


def conv_fwd(x, w_packed, bias, y, k, cin, cout, scale=None, shift=None, act=None, ref=None, mfma=False,
             want_stats=False, bp=None):
    """want_stats: also emit the first stage of the statistics of y (tem_conv3d_fwd_stats) when this launch can; returns
    (partials [N, nblk, cout, 2], nblk) then, else y (and `None` for launches that cannot: use norm_stats).
    bp: Byproducts of this call (tem_conv3d_fwd_ex), not together with want_stats."""
    if bp is not None and want_stats:
        raise ValueError("conv_fwd: by-products and want_stats exclude each other")
    _req_cuda(x, w_packed, y)
    N, D, H, W, C, x_ld = _act5(x)
    Ny, Dy, Hy, Wy, Cy, y_ld = _act5(y)
    if C != cin or Cy != cout or (N, D, H, W) != (Ny, Dy, Hy, Wy):
        raise ValueError(f"conv_fwd: shape mismatch x{tuple(x.shape)} y{tuple(y.shape)} cin={cin} cout={cout}")
    ref_ld = 0
    if ref is not None:
        ref_ld = _act5(ref)[5]
        _same_st(y, ref)
    lib = _lib.load()
    mode = _mode(mfma, x, y)
    nws = lib.tem_conv3d_fwd_ws(N, D, H, W, cin, cout, k[0], k[1], k[2], mode) if mfma else 0
    ws = _workspace(nws, x.device) if nws else None
    kind = None
    if PROFILER is not None:
        pp = lib.tem_conv3d_fwd_kernel(N, D, H, W, cin, cout, k[0], k[1], k[2], mode) if mfma else 0
        kind = _fwd_tag(mfma, k, cout, pp)
    nblk = lib.tem_conv3d_fwd_stat_blocks(N, D, H, W, cin, cout, k[0], k[1], k[2], mode) if want_stats else 0
    ev0 = _prof_begin(x, kind)
    part = None
    if _cs(x) or _cs(y):
        if want_stats and nblk <= 0:
            raise RuntimeError("conv_fwd: a planar tensor needs the z-reuse kernel (statistics rows expected)")
        if nblk > 0:
            part = torch.empty((N, nblk, cout, 2), dtype=torch.float32, device=x.device)
        _lib.check(lib.tem_conv3d_fwd_ex(_p(x), x_ld, _p(scale), _p(shift), _p(w_packed), _p(bias), _p(y), y_ld, _p(ref),
                                         ref_ld, _p(ws), nws, N, D, H, W, cin, cout, k[0], k[1], k[2], ACT[act], mode,
                                         None, None, _p(part), nblk if part is not None else 0, _cs(x), _cs(y),
                                         bp.ref() if bp is not None else None, _stream(x)), "tem_conv3d_fwd_ex")
    elif nblk > 0:
        part = torch.empty((N, nblk, cout, 2), dtype=torch.float32, device=x.device)
        _lib.check(lib.tem_conv3d_fwd_stats(_p(x), x_ld, _p(scale), _p(shift), _p(w_packed), _p(bias), _p(y), y_ld,
                                            _p(ref), ref_ld, _p(ws), nws, N, D, H, W, cin, cout, k[0], k[1], k[2],
                                            ACT[act], mode, _p(part), nblk, _stream(x)), "tem_conv3d_fwd_stats")
    elif bp is not None:
        _lib.check(lib.tem_conv3d_fwd_ex(_p(x), x_ld, _p(scale), _p(shift), _p(w_packed), _p(bias), _p(y), y_ld, _p(ref),
                                         ref_ld, _p(ws), nws, N, D, H, W, cin, cout, k[0], k[1], k[2], ACT[act], mode,
                                         None, None, None, 0, 0, 0, bp.ref(), _stream(x)), "tem_conv3d_fwd_ex")
    else:
        _lib.check(lib.tem_conv3d_fwd(_p(x), x_ld, _p(scale), _p(shift), _p(w_packed), _p(bias), _p(y), y_ld, _p(ref),
                                      ref_ld, _p(ws), nws, N, D, H, W, cin, cout, k[0], k[1], k[2], ACT[act], mode,
                                      _stream(x)), "tem_conv3d_fwd")
    if ev0 is not None:
        _prof_end(x, ev0, (kind, f"{N}x{D}x{H}x{W} {cin}->{cout}"), 2.0 * N * D * H * W * cin * cout * k[0] * k[1] * k[2])
    if want_stats:
        return None if part is None else (part, int(nblk))
    return y


_ws_cache = {}


def _workspace(nbytes: int, device) -> torch.Tensor:
    """A per-(device, stream) scratch buffer (grown on demand, reused across calls on the same stream; kernels of
    different streams may run concurrently, so they never share one)."""
    device = torch.device(device)
    dev = device.index if device.index is not None else torch.cuda.current_device()
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def conv_wgrad_gnorm_ok(k, cin, cout, mfma) -> bool:
    return bool(_lib.load().tem_conv3d_wgrad_gnorm_ok(cin, cout, k[0], k[1], k[2], int(mfma)))


def conv_wgrad_gnorm(x, g, y, coef, k, cin, cout, dw_out, db_out=None, scale=None, shift=None):
    """First-layer weight gradient with the backward of the norm behind this conv's ReLU applied to g on load
    (tem_conv3d_wgrad_gnorm): g raw data gradient, y this conv's output, coef from norm_bwd_coef."""
    _req_cuda(x, g, y, coef, dw_out)
    N, D, H, W, C, x_ld = _act5(x)
    g_ld, y_ld = _act5(g)[5], _act5(y)[5]
    _same_st(g, y)
    lib = _lib.load()
    nws = lib.tem_conv3d_wgrad_ws(N, D, H, W, cin, cout, k[0], k[1], k[2], 0)
    ws = _workspace(nws, x.device)
    kind = _wgrad_tag(0, k, cout) if PROFILER is not None else None
    ev0 = _prof_begin(x, kind)
    if _st(x) or _st(g):
        _lib.check(lib.tem_conv3d_wgrad_gnorm_st(_p(x), x_ld, _p(scale), _p(shift), _p(g), g_ld, _p(y), y_ld, _p(coef), _p(dw_out),
                                                 _p(db_out), _p(ws), nws, N, D, H, W, cin, cout, k[0], k[1], k[2], 1, _st(x), _st(g),
                                                 _stream(x)), "tem_conv3d_wgrad_gnorm_st")
    else:
        _lib.check(lib.tem_conv3d_wgrad_gnorm(_p(x), x_ld, _p(scale), _p(shift), _p(g), g_ld, _p(y), y_ld, _p(coef), _p(dw_out),
                                              _p(db_out), _p(ws), nws, N, D, H, W, cin, cout, k[0], k[1], k[2], 1, _stream(x)),
                   "tem_conv3d_wgrad_gnorm")
    if ev0 is not None:
        _prof_end(x, ev0, (kind, f"{N}x{D}x{H}x{W} {cin}->{cout}"), 2.0 * N * D * H * W * cin * cout * k[0] * k[1] * k[2])


def conv_wgrad_sums_ok(x, k, cin, cout, mfma) -> bool:
    N, D, H, W, _, _ = _act5(x)
    return bool(_lib.load().tem_conv3d_wgrad_sums_ok(N, D, H, W, cin, cout, k[0], k[1], k[2], _mode(mfma, x, x)))


def conv_wgrad(x, g, k, cin, cout, dw_out, db_out=None, scale=None, shift=None, mfma=False, sums_from=None, bp=None):
    """sums_from = (weight [state_dict layout], gamma, beta): also return sums[N, cin, 2] = (sum gz, sum gz*xn) of the
    norm in front of this conv (tem_conv3d_wgrad_sums; check conv_wgrad_sums_ok first)."""
    if sums_from is not None:
        return _conv_wgrad_sums(x, g, k, cin, cout, dw_out, db_out, scale, shift, mfma, sums_from, bp)
    return _conv_wgrad(x, g, k, cin, cout, dw_out, db_out, scale, shift, mfma)


def _conv_wgrad_sums(x, g, k, cin, cout, dw_out, db_out, scale, shift, mfma, sums_from, bp=None):
    _req_cuda(x, g, dw_out, db_out)
    w, gamma, beta = sums_from
    N, D, H, W, C, x_ld = _act5(x)
    g_ld = _act5(g)[5]
    lib = _lib.load()
    mode = _mode(mfma, x, g)
    nws = lib.tem_conv3d_wgrad_ws(N, D, H, W, cin, cout, k[0], k[1], k[2], mode)
    ws = _workspace(nws, x.device)
    sums = torch.empty((N, cin, 2), dtype=torch.float32, device=x.device)
    kind = _wgrad_tag(mfma, k, cout) if PROFILER is not None else None
    ev0 = _prof_begin(x, kind)
    _lib.check(lib.tem_conv3d_wgrad_ex(_p(x), x_ld, _p(scale), _p(shift), _p(g), g_ld, _p(w.detach()), _p(gamma), _p(beta),
                                       _p(dw_out), _p(db_out), _p(sums), None, None, _p(ws), nws, N, D, H, W, cin, cout, k[0], k[1],
                                       k[2], mode, _cs(x), bp.ref() if bp is not None else None, _stream(x)), "tem_conv3d_wgrad_ex")
    if ev0 is not None:
        _prof_end(x, ev0, (kind, f"{N}x{D}x{H}x{W} {cin}->{cout}"), 2.0 * N * D * H * W * cin * cout * k[0] * k[1] * k[2])
    return sums


def conv_wgrad_gmax_ok(x, k, cin, cout, mfma) -> bool:
    N, D, H, W, _, _ = _act5(x)
    return bool(_lib.load().tem_conv3d_wgrad_gmax_ok(N, D, H, W, cin, cout, k[0], k[1], k[2], int(mfma)))


def conv_wgrad_gmax(x, g, k, cin, cout, dw_out, db_out, gmax, scale=None, shift=None, mfma=2, sums_from=None, bp=None):
    """conv_wgrad that also leaves the bit pattern of max |g| in `gmax` (int32[1], cleared by the caller) -- the prescale
    of the fp16 two-term data gradient (conv_fwd_gscaled).  sums_from as in conv_wgrad -> sums[N, cin, 2] or None."""
    _req_cuda(x, g, dw_out, gmax)
    N, D, H, W, C, x_ld = _act5(x)
    g_ld = _act5(g)[5]
    lib = _lib.load()
    nws = lib.tem_conv3d_wgrad_ws(N, D, H, W, cin, cout, k[0], k[1], k[2], int(mfma))
    ws = _workspace(nws, x.device)
    w = gamma = beta = sums = None
    if sums_from is not None:
        w, gamma, beta = sums_from
        w = w.detach()
        sums = torch.empty((N, cin, 2), dtype=torch.float32, device=x.device)
    kind = _wgrad_tag(mfma, k, cout) if PROFILER is not None else None
    ev0 = _prof_begin(x, kind)
    _lib.check(lib.tem_conv3d_wgrad_ex(_p(x), x_ld, _p(scale), _p(shift), _p(g), g_ld, _p(w), _p(gamma), _p(beta),
                                       _p(dw_out), _p(db_out), _p(sums), None, _p(gmax), _p(ws), nws, N, D, H, W, cin, cout,
                                       k[0], k[1], k[2], int(mfma), 0, bp.ref() if bp is not None else None, _stream(x)),
               "tem_conv3d_wgrad_ex")
    if ev0 is not None:
        _prof_end(x, ev0, (kind, f"{N}x{D}x{H}x{W} {cin}->{cout}"), 2.0 * N * D * H * W * cin * cout * k[0] * k[1] * k[2])
    return sums


def conv1x1_out_bwd_ok(cin, cout) -> bool:
    return bool(_lib.load().tem_conv1x1_out_bwd_ok(int(cin), int(cout)))


def conv1x1_out_bwd(x, g, w, gx, dw_out, db_out=None, out_amax=None):
    """Backward of the output projection in one pass over its input x (tem_conv1x1_out_bwd): dw [cout, cin, 1, 1, 1]-ordered, db,
    and gx = (x > 0) * (g . w).  x, gx: [N, D, H, W, cin(ld)]; g: [N, D, H, W, cout(ld)]; w: the conv's weight (state_dict)."""
    _req_cuda(x, g, w, gx, dw_out)
    N, D, H, W, cin, x_ld = _act5(x)
    cout, g_ld = _act5(g)[4], _act5(g)[5]
    gx_ld = _act5(gx)[5]
    lib = _lib.load()
    nws = lib.tem_conv1x1_out_bwd_ws(cin, cout)
    ws = _workspace(nws, x.device)
    _same_st(x, gx)
    if _st(x) or _st(g) or out_amax is not None:
        _lib.check(lib.tem_conv1x1_out_bwd_st(_p(x), x_ld, _p(g), g_ld, _p(w.detach()), _p(gx), gx_ld, _p(dw_out), _p(db_out),
                                              _p(ws), nws, N * D * H * W, cin, cout, _p(out_amax), _st(x), _st(g), _stream(x)),
                   "tem_conv1x1_out_bwd_st")
        return gx
    _lib.check(lib.tem_conv1x1_out_bwd(_p(x), x_ld, _p(g), g_ld, _p(w.detach()), _p(gx), gx_ld, _p(dw_out), _p(db_out), _p(ws), nws,
                                       N * D * H * W, cin, cout, _stream(x)), "tem_conv1x1_out_bwd")
    return gx


def conv_wgrad_gscaled_ok(x, k, cin, cout) -> bool:
    N, D, H, W, _, _ = _act5(x)
    return bool(_lib.load().tem_conv3d_wgrad_gscaled_ok(N, D, H, W, cin, cout, k[0], k[1], k[2]))


def conv_wgrad_cs_ok(x, k, cin, cout, x_cs) -> bool:
    """does the weight gradient honour the chunk stride `x_cs` (elements) of a Planar input of x's size and element type?"""
    N, D, H, W, _, _ = _act5(x)
    return bool(_lib.load().tem_conv3d_wgrad_cs_ok(N, D, H, W, cin, cout, k[0], k[1], k[2], _ST[x.dtype], int(x_cs)))


def absmax(x, amax=None):
    """bit pattern of max |x| of an activation tensor [N, D, H, W, C(ld)] -> int32[1] on the device (tem_absmax);
    `amax` (cleared by the caller) accumulates when given."""
    _req_cuda(x)
    N, D, H, W, C, ld = _act5(x)
    if amax is None:
        amax = torch.zeros(1, dtype=torch.int32, device=x.device)
    _lib.check(_lib.load().tem_absmax(_p(x), ld, C, N * D * H * W, _p(amax), _stream(x)), "tem_absmax")
    return amax


class Byproducts:
    """Optional by-products of ONE library call, passed explicitly (include/tem_hip.h: TemByproducts):
      out_amax  = int32[1] (cleared by the caller): bit pattern of max |output| of the tensor the call writes;
      norm_coef = (groups, mean, rstd, coef [N, C, 4]): a weight gradient that delivers the norm sums (sums_from=...) also
                  finishes them into what norm_bwd_coef(sums=...) returns for a norm without affine parameters;
      norm_sums = (x, groups, mean, rstd, part [N, nblk, C, 2]): a data gradient on the split-K z-reuse kernel also writes
                  the first stage of the backward of the norm whose input is x.
    After the call `amax` / `coef` / `sums` tell which of them it delivered (the caller runs the separate stage otherwise)."""

    def __init__(self, out_amax=None, norm_coef=None, norm_sums=None):
        c = _lib.Byproducts()
        self._keep = (out_amax, norm_coef, norm_sums)
        c.out_amax = _p(out_amax)
        if norm_coef is not None:
            groups, mean, rstd, coef = norm_coef
            c.coef_G, c.coef_mean, c.coef_rstd, c.coef = int(groups), _p(mean), _p(rstd), _p(coef)
        if norm_sums is not None:
            x, groups, mean, rstd, part = norm_sums
            c.sums_x, c.sums_x_ld, c.sums_mean, c.sums_rstd = _p(x), _act5(x)[5], _p(mean), _p(rstd)
            c.sums_G, c.sums_part, c.sums_nblk = int(groups), _p(part), part.shape[1]
        self.c = c

    def ref(self):
        import ctypes
        return ctypes.cast(ctypes.pointer(self.c), ctypes.c_void_p)

    amax = property(lambda self: bool(self.c.delivered & _lib.BP_OUT_AMAX))
    coef = property(lambda self: bool(self.c.delivered & _lib.BP_NORM_COEF))
    sums = property(lambda self: bool(self.c.delivered & _lib.BP_NORM_SUMS))


def conv_wgrad_gscaled(x, g, k, cin, cout, dw_out, db_out, amax, scale=None, shift=None, sums_from=None, bp=None):
    """Weight gradient in the fp16 2x1 arithmetic (tem_conv3d_wgrad_gscaled): x^ two fp16 terms, g one fp16 term prescaled
    from amax = int32[1] with the bit pattern of max |g| (absmax or a producer of g).  sums_from as in conv_wgrad."""
    _req_cuda(x, g, dw_out, amax)
    N, D, H, W, C, x_ld = _act5(x)
    g_ld = _act5(g)[5]
    lib = _lib.load()
    nws = lib.tem_conv3d_wgrad_ws(N, D, H, W, cin, cout, k[0], k[1], k[2], 8)
    ws = _workspace(nws, x.device)
    w = gamma = beta = sums = None
    if sums_from is not None:
        w, gamma, beta = sums_from
        w = w.detach()
        sums = torch.empty((N, cin, 2), dtype=torch.float32, device=x.device)
    kind = _wgrad_tag(8, k, cout) if PROFILER is not None else None
    ev0 = _prof_begin(x, kind)
    _lib.check(lib.tem_conv3d_wgrad_ex(_p(x), x_ld, _p(scale), _p(shift), _p(g), g_ld, _p(w), _p(gamma), _p(beta),
                                       _p(dw_out), _p(db_out), _p(sums), _p(amax), None, _p(ws), nws, N, D, H, W, cin, cout,
                                       k[0], k[1], k[2], 8, 0, bp.ref() if bp is not None else None, _stream(x)), "tem_conv3d_wgrad_ex")
    if ev0 is not None:
        _prof_end(x, ev0, (kind, f"{N}x{D}x{H}x{W} {cin}->{cout}"), 2.0 * N * D * H * W * cin * cout * k[0] * k[1] * k[2])
    return sums


def conv_fwd_gscaled(x, w_packed, y, k, cin, cout, amax, ref=None, bp=None):
    """Data gradient with fp32-class products (tem_conv3d_fwd_gscaled): x an unnormalised gradient, w_packed =
    pack_weights(w, transpose=True, mfma=4), amax = int32[1] holding the bit pattern of max |x| (conv_wgrad_gmax)."""
    _req_cuda(x, w_packed, y, amax)
    N, D, H, W, C, x_ld = _act5(x)
    Ny, Dy, Hy, Wy, Cy, y_ld = _act5(y)
    if C != cin or Cy != cout or (N, D, H, W) != (Ny, Dy, Hy, Wy):
        raise ValueError(f"conv_fwd_gscaled: shape mismatch x{tuple(x.shape)} y{tuple(y.shape)} cin={cin} cout={cout}")
    ref_ld = _act5(ref)[5] if ref is not None else 0
    lib = _lib.load()
    nws = lib.tem_conv3d_fwd_ws(N, D, H, W, cin, cout, k[0], k[1], k[2], 1)
    ws = _workspace(nws, x.device) if nws else None
    kind = _fwd_tag(4, k, cout, 3) if PROFILER is not None else None
    ev0 = _prof_begin(x, kind)
    if bp is not None:
        _lib.check(lib.tem_conv3d_fwd_ex(_p(x), x_ld, None, None, _p(w_packed), None, _p(y), y_ld, _p(ref), ref_ld, _p(ws), nws,
                                         N, D, H, W, cin, cout, k[0], k[1], k[2], ACT[None], 4, _p(amax), None, None, 0, 0, 0,
                                         bp.ref(), _stream(x)), "tem_conv3d_fwd_ex")
    else:
        _lib.check(lib.tem_conv3d_fwd_gscaled(_p(x), x_ld, _p(w_packed), _p(y), y_ld, _p(ref), ref_ld, _p(amax), _p(ws), nws,
                                              N, D, H, W, cin, cout, k[0], k[1], k[2], _stream(x)), "tem_conv3d_fwd_gscaled")
    if ev0 is not None:
        _prof_end(x, ev0, (kind, f"{N}x{D}x{H}x{W} {cin}->{cout}"), 2.0 * N * D * H * W * cin * cout * k[0] * k[1] * k[2])
    return y


def conv_fwd_refnorm(x, w_packed, y, k, cin, cout, ref, coef, mfma, bp=None):
    """Data gradient that lands behind a ReLU + norm (tem_conv3d_fwd_refnorm): y = ref > 0 ? a*conv(x) - m1 - (ref - mean)*m2r
    : 0, coef [N, cout, 4] from norm_bwd_coef.  Only where conv_fwd_family(...) == 3."""
    _req_cuda(x, w_packed, y, ref, coef)
    N, D, H, W, C, x_ld = _act5(x)
    Ny, Dy, Hy, Wy, Cy, y_ld = _act5(y)
    if C != cin or Cy != cout or (N, D, H, W) != (Ny, Dy, Hy, Wy) or tuple(coef.shape) != (N, cout, 4) or \
            not coef.is_contiguous():
        raise ValueError(f"conv_fwd_refnorm: shape mismatch x{tuple(x.shape)} y{tuple(y.shape)} coef{tuple(coef.shape)}")
    ref_ld = _act5(ref)[5]
    _same_st(x, y, ref)
    lib = _lib.load()
    nws = lib.tem_conv3d_fwd_ws(N, D, H, W, cin, cout, k[0], k[1], k[2], 1)
    ws = _workspace(nws, x.device) if nws else None
    kind = _fwd_tag(mfma, k, cout, 3) if PROFILER is not None else None
    ev0 = _prof_begin(x, kind)
    if bp is not None:
        _lib.check(lib.tem_conv3d_fwd_ex(_p(x), x_ld, None, None, _p(w_packed), None, _p(y), y_ld, _p(ref), ref_ld, _p(ws), nws,
                                         N, D, H, W, cin, cout, k[0], k[1], k[2], ACT[None], _mode(mfma, x, y), None, _p(coef),
                                         None, 0, 0, 0, bp.ref(), _stream(x)), "tem_conv3d_fwd_ex")
    else:
        _lib.check(lib.tem_conv3d_fwd_refnorm(_p(x), x_ld, _p(w_packed), _p(y), y_ld, _p(ref), ref_ld, _p(coef), _p(ws), nws,
                                              N, D, H, W, cin, cout, k[0], k[1], k[2], _mode(mfma, x, y), _stream(x)),
                   "tem_conv3d_fwd_refnorm")
    if ev0 is not None:
        _prof_end(x, ev0, (kind, f"{N}x{D}x{H}x{W} {cin}->{cout}"), 2.0 * N * D * H * W * cin * cout * k[0] * k[1] * k[2])
    return y


def conv_fwd_stat_blocks(x, k, cin, cout, mfma) -> int:
    """tem_conv3d_fwd_stat_blocks: statistics partial rows per sample this launch writes (0: it cannot)"""
    N, D, H, W, _, _ = _act5(x)
    return int(_lib.load().tem_conv3d_fwd_stat_blocks(N, D, H, W, cin, cout, k[0], k[1], k[2], _mode(mfma, x, x)))


def conv_fwd_family(x, k, cin, cout, mfma) -> int:
    """tem_conv3d_fwd_kernel: 0 patch / other kernels, 1 / 2 ping-pong teams, 3 z-reuse teams, 4 z-reuse teams with split-K"""
    N, D, H, W, _, _ = _act5(x)
    return int(_lib.load().tem_conv3d_fwd_kernel(N, D, H, W, cin, cout, k[0], k[1], k[2], _mode(mfma, x, x)))


def _conv_wgrad(x, g, k, cin, cout, dw_out, db_out=None, scale=None, shift=None, mfma=False):
    """dw_out: flat [ntaps*cin*cout] in the reference's [Cout,Cin,kd,kh,kw] order; db_out: [cout]."""
    _req_cuda(x, g, dw_out)
    N, D, H, W, C, x_ld = _act5(x)
    _, _, _, _, Cg, g_ld = _act5(g)
    if C != cin or Cg != cout:
        raise ValueError("conv_wgrad: channel mismatch")
    lib = _lib.load()
    mode = _mode(mfma, x, g)
    nws = lib.tem_conv3d_wgrad_ws(N, D, H, W, cin, cout, k[0], k[1], k[2], mode)
    ntaps = k[0] * k[1] * k[2]
    ws = _workspace(nws, x.device)
    kind = _wgrad_tag(mfma, k, cout) if PROFILER is not None else None
    ev0 = _prof_begin(x, kind)
    if _cs(x):
        _lib.check(lib.tem_conv3d_wgrad_ex(_p(x), x_ld, _p(scale), _p(shift), _p(g), g_ld, None, None, None, _p(dw_out), _p(db_out),
                                           None, None, None, _p(ws), nws, N, D, H, W, cin, cout, k[0], k[1], k[2], mode, _cs(x), None,
                                           _stream(x)), "tem_conv3d_wgrad_ex")
    else:
        _lib.check(lib.tem_conv3d_wgrad(_p(x), x_ld, _p(scale), _p(shift), _p(g), g_ld, _p(dw_out), _p(db_out), _p(ws), nws,
                                        N, D, H, W, cin, cout, k[0], k[1], k[2], mode, 1, _stream(x)),
                   "tem_conv3d_wgrad")
    if ev0 is not None:
        _prof_end(x, ev0, (kind, f"{N}x{D}x{H}x{W} {cin}->{cout}"), 2.0 * N * D * H * W * cin * cout * k[0] * k[1] * k[2])
    return dw_out


# ------------------------------------------------------------------ norm ----
def norm_stats(x, groups: int, gamma=None, beta=None, eps: float = 1e-5):
    """-> (mean[N,G], rstd[N,G], scale[N,C], shift[N,C])"""
    _req_cuda(x)
    N, D, H, W, C, ld = _act5(x)
    dev = x.device
    mean = torch.empty((N, groups), dtype=torch.float32, device=dev)
    rstd = torch.empty((N, groups), dtype=torch.float32, device=dev)
    scale = torch.empty((N, C), dtype=torch.float32, device=dev)
    shift = torch.empty((N, C), dtype=torch.float32, device=dev)
    lib = _lib.load()
    V = D * H * W
    nws = lib.tem_norm_ws(N, V, C)
    ws = _workspace(nws, dev)
    if _st(x):
        _lib.check(lib.tem_norm_stats_st(_p(x), ld, N, V, C, groups, _p(gamma), _p(beta), eps, _p(mean), _p(rstd), _p(scale),
                                         _p(shift), _p(ws), nws, _st(x), _stream(x)), "tem_norm_stats_st")
        return mean, rstd, scale, shift
    _lib.check(lib.tem_norm_stats(_p(x), ld, N, V, C, groups, _p(gamma), _p(beta), eps, _p(mean), _p(rstd), _p(scale),
                                  _p(shift), _p(ws), nws, _stream(x)), "tem_norm_stats")
    return mean, rstd, scale, shift


def norm_stats_from_partials(part, rows: int, voxels: int, C: int, groups: int, gamma=None, beta=None,
                             eps: float = 1e-5):
    """Second stage of norm_stats on partial sums written by conv_fwd(want_stats=True).  part: [N, nblk, C, 2];
    rows = N for per-sample statistics, 1 for BatchNorm (then every sample's blocks merge into one row)."""
    _req_cuda(part)
    N, nblk = part.shape[0], part.shape[1]
    if rows == 1:
        nblk, voxels = N * nblk, N * voxels
    elif rows != N:
        raise ValueError("norm_stats_from_partials: rows must be N or 1")
    dev = part.device
    mean = torch.empty((rows, groups), dtype=torch.float32, device=dev)
    rstd = torch.empty((rows, groups), dtype=torch.float32, device=dev)
    scale = torch.empty((rows, C), dtype=torch.float32, device=dev)
    shift = torch.empty((rows, C), dtype=torch.float32, device=dev)
    lib = _lib.load()
    _lib.check(lib.tem_norm_finalize_partials(_p(part), nblk, rows, voxels, C, groups, _p(gamma), _p(beta), eps,
                                              _p(mean), _p(rstd), _p(scale), _p(shift), _stream(part)),
               "tem_norm_finalize_partials")
    return mean, rstd, scale, shift


def norm_bwd_coef(gy, x, groups, gamma, mean, rstd, dgamma=None, dbeta=None, sums=None):
    """Reduction stage of norm_bwd only -> coef[N, C, 4] = (a, m1, m2r, mean) per (sample, channel)."""
    _req_cuda(gy, x)
    if isinstance(x, Planar) and sums is None:
        # no producer delivered the sums: reduce each dense half on its own (groups never straddle the halves: 32 % (C / G) == 0)
        cg = 64 // groups
        if 32 % cg:
            raise ValueError("norm_bwd_coef: a group straddles the halves of a planar tensor")
        gh = 32 // cg
        sl = lambda t, i, n: None if t is None else t[..., i * n:(i + 1) * n]  # noqa: E731
        return torch.cat([norm_bwd_coef(gy.halves[i], x.halves[i], gh, sl(gamma, i, 32), sl(mean, i, gh).contiguous(),
                                        sl(rstd, i, gh).contiguous(), sl(dgamma, i, 32), sl(dbeta, i, 32)) for i in (0, 1)], dim=1)
    N, D, H, W, C, x_ld = _act5(x)
    gy_ld = _act5(gy)[5]
    if isinstance(x, Planar):   # with the sums given the tensors are not read: any valid leading dimension
        x_ld = gy_ld = C
    lib = _lib.load()
    V = D * H * W
    nws = lib.tem_norm_ws(N, V, C)
    ws = _workspace(nws, x.device)
    coef = torch.empty((N, C, 4), dtype=torch.float32, device=x.device)
    if _st(x) or _st(gy):
        _same_st(gy, x)
        nrow = 0 if sums is None else (sums.shape[1] if sums.dim() == 4 else 1)
        _lib.check(lib.tem_norm_bwd_st(_p(gy), gy_ld, _p(x), x_ld, N, V, C, groups, _p(gamma), _p(mean), _p(rstd), 0, None, C,
                                       _p(dgamma), _p(dbeta), _p(sums), nrow, _p(coef), None, _p(ws), nws, _st(x), _stream(x)),
                   "tem_norm_bwd_st")
        return coef
    if sums is not None and sums.dim() == 4:   # partial rows [N, nblk, C, 2] from a data gradient (Byproducts.norm_sums of the data gradient)
        _lib.check(lib.tem_norm_bwd_from_partials(_p(gy), gy_ld, _p(x), x_ld, N, V, C, groups, _p(gamma), _p(mean), _p(rstd), 0,
                                                  None, C, _p(dgamma), _p(dbeta), _p(sums), sums.shape[1], _p(coef), _p(ws), nws,
                                                  _stream(x)), "tem_norm_bwd_from_partials")
        return coef
    _lib.check(lib.tem_norm_bwd_coef(_p(gy), gy_ld, _p(x), x_ld, N, V, C, groups, _p(gamma), _p(mean), _p(rstd),
                                     _p(dgamma), _p(dbeta), _p(sums), _p(coef), _p(ws), nws, _stream(x)),
               "tem_norm_bwd_coef")
    return coef


def norm_bwd(gy, x, groups, gamma, mean, rstd, relu_mask: bool, gx, dgamma=None, dbeta=None, sums=None, out_amax=None):
    """sums: [N, C, 2] from conv_wgrad(sums_from=...) -- skips the reduction pass over gy and x;
    out_amax: int32[1] (cleared by the caller) that receives the bit pattern of max |gx|"""
    _req_cuda(gy, x, gx)
    N, D, H, W, C, x_ld = _act5(x)
    gy_ld = _act5(gy)[5]
    gx_ld = _act5(gx)[5]
    lib = _lib.load()
    V = D * H * W
    nws = lib.tem_norm_ws(N, V, C)
    ws = _workspace(nws, x.device)
    if _st(x) or _st(gy) or _st(gx) or out_amax is not None:
        _same_st(gy, x, gx)
        nrow = 0 if sums is None else (sums.shape[1] if sums.dim() == 4 else 1)
        _lib.check(lib.tem_norm_bwd_st(_p(gy), gy_ld, _p(x), x_ld, N, V, C, groups, _p(gamma), _p(mean), _p(rstd), int(relu_mask),
                                       _p(gx), gx_ld, _p(dgamma), _p(dbeta), _p(sums), nrow, None, _p(out_amax), _p(ws), nws, _st(x),
                                       _stream(x)), "tem_norm_bwd_st")
        return gx
    if sums is not None and sums.dim() == 4:   # partial rows [N, nblk, C, 2] from a data gradient (Byproducts.norm_sums of the data gradient)
        _lib.check(lib.tem_norm_bwd_from_partials(_p(gy), gy_ld, _p(x), x_ld, N, V, C, groups, _p(gamma), _p(mean), _p(rstd),
                                                  int(relu_mask), _p(gx), gx_ld, _p(dgamma), _p(dbeta), _p(sums), sums.shape[1],
                                                  None, _p(ws), nws, _stream(x)), "tem_norm_bwd_from_partials")
        return gx
    if sums is not None:
        _lib.check(lib.tem_norm_bwd_from_sums(_p(gy), gy_ld, _p(x), x_ld, N, V, C, groups, _p(gamma), _p(mean), _p(rstd),
                                              int(relu_mask), _p(gx), gx_ld, _p(dgamma), _p(dbeta), _p(sums), _p(ws), nws,
                                              _stream(x)), "tem_norm_bwd_from_sums")
        return gx
    _lib.check(lib.tem_norm_bwd(_p(gy), gy_ld, _p(x), x_ld, N, V, C, groups, _p(gamma), _p(mean), _p(rstd),
                                int(relu_mask), _p(gx), gx_ld, _p(dgamma), _p(dbeta), _p(ws), nws, _stream(x)),
               "tem_norm_bwd")
    return gx


# --------------------------------------------------------- pool / upsample ----
def maxpool_fwd(x, y, f, want_stats=False):
    """want_stats: also return (partials [N, nblk, C, 2], nblk) -- the first stage of the statistics of y (what
    conv_fwd(want_stats=True) returns for a conv output; tem_maxpool3d_fwd_stats) -- or None when this channel count cannot."""
    _req_cuda(x, y)
    N, D, H, W, C, x_ld = _act5(x)
    y_ld = _act5(y)[5]
    lib = _lib.load()
    nblk = int(lib.tem_maxpool3d_fwd_stat_blocks(D, H, C, f[0], f[1])) if want_stats else 0
    if _st(x) or _st(y):
        _same_st(x, y)
        part = None
        if nblk > 0 and x_ld % 4 == 0 and y_ld % 4 == 0 and x.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0:
            part = torch.empty((N, nblk, C, 2), dtype=torch.float32, device=x.device)
        _lib.check(lib.tem_maxpool3d_fwd_st(_p(x), x_ld, _p(y), y_ld, N, D, H, W, C, f[0], f[1], f[2], _p(part),
                                            nblk if part is not None else 0, _st(x), _stream(x)), "tem_maxpool3d_fwd_st")
        if want_stats:
            return None if part is None else (part, nblk)
        return y
    if nblk > 0 and x_ld % 4 == 0 and y_ld % 4 == 0 and x.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0:
        part = torch.empty((N, nblk, C, 2), dtype=torch.float32, device=x.device)
        _lib.check(lib.tem_maxpool3d_fwd_stats(_p(x), x_ld, _p(y), y_ld, N, D, H, W, C, f[0], f[1], f[2], _p(part), nblk,
                                               _stream(x)), "tem_maxpool3d_fwd_stats")
        return part, nblk
    _lib.check(lib.tem_maxpool3d_fwd(_p(x), x_ld, _p(y), y_ld, N, D, H, W, C, f[0], f[1], f[2], _stream(x)),
               "tem_maxpool3d_fwd")
    return None if want_stats else y


def maxpool_bwd(gy, x, gx, f, gskip=None, relu_mask=False, gskip_coef=None, gy_coef=None, out_amax=None):
    """gskip_coef: [N, C, 4] view (row stride = multiple of 4 floats) of norm_bwd_coef() -- gskip is then the raw data
    gradient behind that norm and the norm backward is applied on the fly (tem_maxpool3d_bwd_norm)."""
    _req_cuda(gy, x, gx)
    N, D, H, W, C, x_ld = _act5(x)
    gy_ld = _act5(gy)[5]
    gx_ld = _act5(gx)[5]
    gs_ld = _act5(gskip)[5] if gskip is not None else 0
    lib = _lib.load()
    if gy_coef is not None and not gy_coef.is_contiguous():
        raise ValueError("maxpool_bwd: gy_coef must be contiguous")
    if _st(x) or _st(gy) or _st(gx) or out_amax is not None:
        _same_st(gy, x, gx, gskip)
        _lib.check(lib.tem_maxpool3d_bwd_st(_p(gy), gy_ld, _p(x), x_ld, _p(gskip), gs_ld, int(relu_mask), _p(gx), gx_ld,
                                            N, D, H, W, C, f[0], f[1], f[2], _p(gskip_coef),
                                            gskip_coef.stride(0) if gskip_coef is not None else 0, _p(gy_coef), _p(out_amax), _st(x),
                                            _stream(x)), "tem_maxpool3d_bwd_st")
        return gx
    if gskip_coef is not None or gy_coef is not None:
        # gy_coef: dense [N, C, 4] coefficients of the norm whose input is the pooled tensor (gy raw as well)
        if gy_coef is not None and not gy_coef.is_contiguous():
            raise ValueError("maxpool_bwd: gy_coef must be contiguous")
        _lib.check(lib.tem_maxpool3d_bwd_norm(_p(gy), gy_ld, _p(x), x_ld, _p(gskip), gs_ld, int(relu_mask), _p(gx), gx_ld,
                                              N, D, H, W, C, f[0], f[1], f[2], _p(gskip_coef),
                                              gskip_coef.stride(0) if gskip_coef is not None else 0, _p(gy_coef),
                                              _stream(x)), "tem_maxpool3d_bwd_norm")
        return gx
    _lib.check(lib.tem_maxpool3d_bwd(_p(gy), gy_ld, _p(x), x_ld, _p(gskip), gs_ld, int(relu_mask), _p(gx), gx_ld,
                                     N, D, H, W, C, f[0], f[1], f[2], _stream(x)), "tem_maxpool3d_bwd")
    return gx


def upsample_fwd(x, y, f, stats: bool = False):
    """y = interpolate(x, scale_factor=f, trilinear).  stats=True: also return the first stage of y's statistics,
    part [N, D*H, C, 2] (tem_upsample_fwd_stats), or None when the factor-2 kernel does not take the shape -- then
    `upsample_stats` derives them from x."""
    _req_cuda(x, y)
    N, D, H, W, C, x_ld = _act5(x)
    y_ld = _act5(y)[5]
    lib = _lib.load()
    if _st(x) or _st(y):
        _same_st(x, y)
        part = None
        if stats and lib.tem_upsample_fwd_stats_ok(C, f[0], f[1], f[2]) and x_ld % 4 == 0 and y_ld % 4 == 0 and \
                x.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0:
            part = torch.empty((N, D * H, C, 2), dtype=torch.float32, device=x.device)
        _lib.check(lib.tem_upsample_fwd_st(_p(x), x_ld, _p(y), y_ld, N, D, H, W, C, f[0], f[1], f[2], _p(part), _st(x),
                                           _stream(x)), "tem_upsample_fwd_st")
        return part if stats else y
    if stats:
        if not (lib.tem_upsample_fwd_stats_ok(C, f[0], f[1], f[2]) and x_ld % 4 == 0 and y_ld % 4 == 0 and
                x.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0):
            upsample_fwd(x, y, f)
            return None
        part = torch.empty((N, D * H, C, 2), dtype=torch.float32, device=x.device)
        _lib.check(lib.tem_upsample_fwd_stats(_p(x), x_ld, _p(y), y_ld, N, D, H, W, C, f[0], f[1], f[2], _p(part),
                                              _stream(x)), "tem_upsample_fwd_stats")
        return part
    _lib.check(lib.tem_upsample_fwd(_p(x), x_ld, _p(y), y_ld, N, D, H, W, C, f[0], f[1], f[2], _stream(x)),
               "tem_upsample_fwd")
    return y


def upsample_stats_ok(u) -> bool:
    C = u.shape[4]
    cq = C // 4
    return C % 4 == 0 and 0 < cq <= 64 and (cq & (cq - 1)) == 0 and _act5(u)[5] % 4 == 0 and u.data_ptr() % 16 == 0


def upsample_stats(u, f):
    """First stage of the statistics of upsample(u, f), from u alone -> part [N, D*H, C, 2] (tem_upsample_stats)."""
    _req_cuda(u)
    N, D, H, W, C, u_ld = _act5(u)
    part = torch.empty((N, D * H, C, 2), dtype=torch.float32, device=u.device)
    lib = _lib.load()
    if _st(u):
        _lib.check(lib.tem_upsample_stats_st(_p(u), u_ld, N, D, H, W, C, f[0], f[1], f[2], _p(part), _st(u), _stream(u)),
                   "tem_upsample_stats_st")
        return part
    _lib.check(lib.tem_upsample_stats(_p(u), u_ld, N, D, H, W, C, f[0], f[1], f[2], _p(part), _stream(u)),
               "tem_upsample_stats")
    return part


def norm_stats_from_partials2(part_a, part_b, rows: int, voxels: int, groups: int, gamma=None, beta=None,
                              eps: float = 1e-5):
    """norm_stats of a channel-concatenated tensor: channels [0, CA) summarised by part_a [N, nblkA, CA, 2], the rest by
    part_b [N, nblkB, CB, 2] (tem_norm_finalize_partials2).  rows = N, or 1 for BatchNorm."""
    _req_cuda(part_a, part_b)
    N, nba, ca = part_a.shape[0], part_a.shape[1], part_a.shape[2]
    nbb, cb = part_b.shape[1], part_b.shape[2]
    C = ca + cb
    if rows == 1:
        nba, nbb, voxels = N * nba, N * nbb, N * voxels
    elif rows != N:
        raise ValueError("norm_stats_from_partials2: rows must be N or 1")
    dev = part_a.device
    mean = torch.empty((rows, groups), dtype=torch.float32, device=dev)
    rstd = torch.empty((rows, groups), dtype=torch.float32, device=dev)
    scale = torch.empty((rows, C), dtype=torch.float32, device=dev)
    shift = torch.empty((rows, C), dtype=torch.float32, device=dev)
    lib = _lib.load()
    _lib.check(lib.tem_norm_finalize_partials2(_p(part_a), nba, ca, _p(part_b), nbb, rows, voxels, C, groups, _p(gamma),
                                               _p(beta), eps, _p(mean), _p(rstd), _p(scale), _p(shift), _stream(part_a)),
               "tem_norm_finalize_partials2")
    return mean, rstd, scale, shift


def upsample_bwd(gy, gx, f, norm=None):
    """gx has the low-resolution shape; gy = gx's shape scaled by f.
    norm = (u, coef[N, C, 4] view): gy is the raw data gradient behind a norm whose input was upsample(u)."""
    _req_cuda(gy, gx)
    N, D, H, W, C, gx_ld = _act5(gx)
    gy_ld = _act5(gy)[5]
    lib = _lib.load()
    if _st(gy) or _st(gx):
        u, coef = norm if norm is not None else (None, None)
        _same_st(gy, gx, u)
        _lib.check(lib.tem_upsample_bwd_st(_p(gy), gy_ld, _p(gx), gx_ld, N, D, H, W, C, f[0], f[1], f[2], _p(u),
                                           _act5(u)[5] if u is not None else 0, _p(coef), coef.stride(0) if coef is not None else 0,
                                           _st(gy), _stream(gx)), "tem_upsample_bwd_st")
        return gx
    if norm is not None:
        u, coef = norm
        _lib.check(lib.tem_upsample_bwd_norm(_p(gy), gy_ld, _p(gx), gx_ld, N, D, H, W, C, f[0], f[1], f[2], _p(u),
                                             _act5(u)[5], _p(coef), coef.stride(0), _stream(gx)), "tem_upsample_bwd_norm")
        return gx
    _lib.check(lib.tem_upsample_bwd(_p(gy), gy_ld, _p(gx), gx_ld, N, D, H, W, C, f[0], f[1], f[2], _stream(gx)),
               "tem_upsample_bwd")
    return gx


def act_bwd(gy, y, act: str):
    _req_cuda(gy, y)
    gy = gy.contiguous()
    y = y.contiguous()
    gx = torch.empty_like(gy)
    lib = _lib.load()
    _lib.check(lib.tem_act_bwd(_p(gy), _p(y), _p(gx), gy.numel(), ACT[act], _stream(gy)), "tem_act_bwd")
    return gx


# ------------------------------------------------------------------ dice ----
def _ncv_strides(t: torch.Tensor):
    """(sn, sc, sv, N, C, V) for a logical [N, C, *spatial] tensor whose spatial dims are
    jointly contiguous up to a common voxel stride; returns None if not expressible."""
    N, C = t.shape[:2]
    sp = tuple(t.shape[2:])
    V = 1
    for s in sp:
        V *= s
    st = t.stride()
    sv = st[-1] if len(sp) > 0 else 1
    # check spatial dims collapse to a single stride sv
    expect = sv
    for d in range(len(sp) - 1, -1, -1):
        if sp[d] != 1 and st[2 + d] != expect:
            return None
        expect *= sp[d]
    return st[0], st[1], sv, N, C, V


def dice_sums(p, t, mask=None) -> torch.Tensor:
    """-> double[C, 3] = (sum p*t, sum p*p, sum t*t) with optional multiplicative mask."""
    _req_cuda(p, t)
    ps = _ncv_strides(p)
    if ps is None:
        p = p.contiguous()
        ps = _ncv_strides(p)
    ts = _ncv_strides(t)
    if ts is None:
        t = t.contiguous()
        ts = _ncv_strides(t)
    if mask is not None:
        ms = _ncv_strides(mask)
        if ms is None or ms[:3] != ts[:3]:
            raise ValueError("dice: mask must share the target's strides")
    N, C, V = ps[3:]
    sums = torch.empty((C, 3), dtype=torch.float64, device=p.device)
    lib = _lib.load()
    nws = lib.tem_dice_ws(N, V, C)
    ws = _workspace(nws, p.device)
    _lib.check(lib.tem_dice_sums(_p(p), ps[0], ps[1], ps[2], _p(t), ts[0], ts[1], ts[2], _p(mask), N, C, V,
                                 _p(sums), _p(ws), nws, _stream(p)), "tem_dice_sums")
    return sums, p, t


REDUCE = {None: 0, "sum": 1, "mean": 2, "max": 3, "min": 4}
DICE_LOGITS, DICE_BCE = 1, 2


def dice_sums2(p, t, flags: int):
    """dice_sums for the logits / BCE members of the family (tem_dice_sums2): double[C, 3] or, with DICE_BCE, [C, 4]
    (4th column: summed binary cross entropy of the channel)."""
    _req_cuda(p, t)
    ps = _ncv_strides(p)
    if ps is None:
        p = p.contiguous()
        ps = _ncv_strides(p)
    ts = _ncv_strides(t)
    if ts is None:
        t = t.contiguous()
        ts = _ncv_strides(t)
    N, C, V = ps[3:]
    ncol = 4 if flags & DICE_BCE else 3
    sums = torch.empty((C, ncol), dtype=torch.float64, device=p.device)
    lib = _lib.load()
    nws = lib.tem_dice_ws(N, V, C)
    ws = _workspace(nws, p.device)
    _lib.check(lib.tem_dice_sums2(_p(p), ps[0], ps[1], ps[2], _p(t), ts[0], ts[1], ts[2], N, C, V, _p(sums), _p(ws), nws,
                                  int(flags), _stream(p)), "tem_dice_sums2")
    return sums, p, t


def dice_finalize2(sums, eps, channelwise, invert, reduce):
    C, ncol = sums.shape
    dev = sums.device
    n_out = C if (channelwise and reduce is None) else 1
    out = torch.empty((n_out,), dtype=torch.float32, device=dev)
    ca = torch.empty((C,), dtype=torch.float32, device=dev)
    cb = torch.empty((C,), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().tem_dice_finalize2(_p(sums), ncol, C, float(eps), int(channelwise), int(invert), REDUCE[reduce],
                                              _p(out), _p(ca), _p(cb), _stream(sums)), "tem_dice_finalize2")
    return out, ca, cb


def dice_grad2(p, t, ca, cb, gout, gout_per_channel, channels_last: bool, flags: int, w_dice: float, w_bce: float):
    ps = _ncv_strides(p)
    ts = _ncv_strides(t)
    N, C, V = ps[3:]
    if channels_last:
        gp_phys = torch.empty((N,) + tuple(p.shape[2:]) + (C,), dtype=torch.float32, device=p.device)
        gp = gp_phys.permute(0, gp_phys.dim() - 1, *range(1, gp_phys.dim() - 1))
        gs = (V * C, 1, C)
    else:
        gp = torch.empty(p.shape, dtype=torch.float32, device=p.device)
        gs = (C * V, V, 1)
    _lib.check(_lib.load().tem_dice_grad2(_p(p), ps[0], ps[1], ps[2], _p(t), ts[0], ts[1], ts[2], _p(ca), _p(cb), _p(gout),
                                          int(gout_per_channel), _p(gp), gs[0], gs[1], gs[2], N, C, V, int(flags),
                                          float(w_dice), float(w_bce), _stream(p)), "tem_dice_grad2")
    return gp


def dice_finalize(sums, eps, channelwise, invert, reduce):
    C = sums.shape[0]
    dev = sums.device
    n_out = C if (channelwise and reduce is None) else 1
    out = torch.empty((n_out,), dtype=torch.float32, device=dev)
    ca = torch.empty((C,), dtype=torch.float32, device=dev)
    cb = torch.empty((C,), dtype=torch.float32, device=dev)
    lib = _lib.load()
    _lib.check(lib.tem_dice_finalize(_p(sums), C, float(eps), int(channelwise), int(invert), REDUCE[reduce], _p(out),
                                     _p(ca), _p(cb), _stream(sums)), "tem_dice_finalize")
    return out, ca, cb


def dice_grad(p, t, mask, ca, cb, gout, gout_per_channel, channels_last: bool):
    """d out / d p, laid out like p (channels_last_3d memory format when p has it)."""
    ps = _ncv_strides(p)
    ts = _ncv_strides(t)
    N, C, V = ps[3:]
    if channels_last:
        # physical [N, V, C]
        gp_phys = torch.empty((N,) + tuple(p.shape[2:]) + (C,), dtype=torch.float32, device=p.device)
        gp = gp_phys.permute(0, gp_phys.dim() - 1, *range(1, gp_phys.dim() - 1))
        gs = (V * C, 1, C)
    else:
        gp = torch.empty(p.shape, dtype=torch.float32, device=p.device)
        gs = (C * V, V, 1)
    lib = _lib.load()
    _lib.check(lib.tem_dice_grad(_p(p), ps[0], ps[1], ps[2], _p(t), ts[0], ts[1], ts[2], _p(mask), _p(ca), _p(cb),
                                 _p(gout), int(gout_per_channel), _p(gp), gs[0], gs[1], gs[2], N, C, V, _stream(p)),
               "tem_dice_grad")
    return gp


# ------------------------------------------------------------- optimizer ----
def bump_versions(tensors):
    """The kernels below write parameters through raw pointers, which autograd's version counters do not see;
    consumers that cache derived data per `tensor._version` (the engine's packed weight fragments) must be
    told.  ONE host call for all of them, no device work.  (`_increment_version` takes an iterable of tensors: handing
    it a single tensor makes it iterate over the tensor's ROWS -- 512 view objects for a [512, 512, 3, 3, 3] weight;
    46 such calls were 6.5 ms of host time per optimizer step.)"""
    tensors = list(tensors)
    inc = getattr(torch._C, "_increment_version", None)
    if inc is not None:
        inc(tensors)
        return
    for t in tensors:  # pragma: no cover -- very old torch: a no-op in-place op bumps the counter
        t.add_(0)


def adamw_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    _req_cuda(param, grad, exp_avg, exp_avg_sq)
    lib = _lib.load()
    _lib.check(lib.tem_adamw_step(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), lr, beta1, beta2,
                                  eps, weight_decay, int(step), grad_scale, _stream(param)), "tem_adamw_step")


def adamw_hyper(host_buf, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    """Fill the 12-float HOST tensor that `adamw_step_dev` reads (after a copy to the device): tem_adamw_hyper."""
    assert not host_buf.is_cuda and host_buf.dtype == torch.float32 and host_buf.numel() >= 12
    lib = _lib.load()
    _lib.check(lib.tem_adamw_hyper(ctypes.c_void_p(host_buf.data_ptr()), lr, beta1, beta2, eps, weight_decay, int(step),
                                   grad_scale), "tem_adamw_hyper")


def adamw_step_dev(param, grad, exp_avg, exp_avg_sq, hyper):
    """adamw_step with lr / bias corrections read from the device tensor `hyper` (HIP-graph capture)."""
    _req_cuda(param, grad, exp_avg, exp_avg_sq, hyper)
    lib = _lib.load()
    _lib.check(lib.tem_adamw_step_dev(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), _p(hyper),
                                      _stream(param)), "tem_adamw_step_dev")


def adamw_step_tab(param, grad, exp_avg, exp_avg_sq, table, sstate):
    """adamw_step with its scalars from the row of `table` that the device-side step count in `sstate` selects; skipped
    when sstate's overflow flag is up (tem_adamw_step_tab)."""
    _req_cuda(param, grad, exp_avg, exp_avg_sq, table, sstate)
    lib = _lib.load()
    _lib.check(lib.tem_adamw_step_tab(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), _p(table),
                                      _p(sstate), _stream(param)), "tem_adamw_step_tab")


def amp_unscale_dev(grad, sstate):
    _req_cuda(grad, sstate)
    lib = _lib.load()
    _lib.check(lib.tem_amp_unscale_dev(_p(grad), grad.numel(), _p(sstate), _stream(grad)), "tem_amp_unscale_dev")


def amp_update_dev(sstate, growth, backoff, interval):
    _req_cuda(sstate)
    lib = _lib.load()
    _lib.check(lib.tem_amp_update_dev(_p(sstate), float(growth), float(backoff), int(interval), _stream(sstate)),
               "tem_amp_update_dev")


def amp_unscale(grad, inv_scale, found_inf):
    """grad *= inv_scale in place; found_inf[0] = 1 if any element is not finite (GradScaler.unscale_)."""
    _req_cuda(grad, found_inf)
    lib = _lib.load()
    _lib.check(lib.tem_amp_unscale(_p(grad), grad.numel(), float(inv_scale), _p(found_inf), _stream(grad)),
               "tem_amp_unscale")


def ema_update(theta_k, theta_q, momentum):
    _req_cuda(theta_k, theta_q)
    lib = _lib.load()
    _lib.check(lib.tem_ema_update(_p(theta_k), _p(theta_q), theta_k.numel(), momentum, _stream(theta_k)),
               "tem_ema_update")


# ---------------------------------------------------------------- labels ----
BOUNDARY_MODES = {"thick": 0, "inner": 1, "outer": 2}


def boundary_target(labels: torch.Tensor, add_binary_target: bool, mode: str = "thick") -> torch.Tensor:
    _req_cuda(labels)
    if mode not in BOUNDARY_MODES:
        raise NotImplementedError(f"find_boundaries mode '{mode}': the MI355X kernel has {sorted(BOUNDARY_MODES)} "
                                  "('subpixel' returns a 2n-1 grid, which cannot be a training target)")
    labels = labels.to(torch.int64).contiguous()
    sp = tuple(labels.shape)
    D, H, W = (1,) * (3 - len(sp)) + sp
    nch = 2 if add_binary_target else 1
    out = torch.empty((nch,) + sp, dtype=torch.float32, device=labels.device)
    lib = _lib.load()
    _lib.check(lib.tem_boundary_target_mode(_p(labels), _p(out), D, H, W, int(add_binary_target),
                                            BOUNDARY_MODES[mode], _stream(labels)), "tem_boundary_target_mode")
    return out


def affinity_target(labels, offsets, ignore_label=None, add_binary_target=False, add_mask=False,
                    include_ignore_transitions=False) -> torch.Tensor:
    _req_cuda(labels)
    labels = labels.to(torch.int64).contiguous()
    sp = tuple(labels.shape)
    nd = len(sp)
    D, H, W = (1,) * (3 - nd) + sp
    offs = []
    for o in offsets:
        o = [int(v) for v in o]
        if len(o) != nd:
            raise ValueError("offset dimensionality does not match the labels")
        offs.extend([0] * (3 - nd) + o)
    n_off = len(offsets)
    arr = (ctypes.c_int * len(offs))(*offs)
    cb = 1 if add_binary_target else 0
    nch = (n_off + cb) * (2 if add_mask else 1)
    out = torch.empty((nch,) + sp, dtype=torch.float32, device=labels.device)
    lib = _lib.load()
    _lib.check(lib.tem_affinity_target(_p(labels), _p(out), D, H, W, arr, n_off, int(ignore_label is not None),
                                       int(ignore_label or 0), int(add_binary_target), int(add_mask),
                                       int(include_ignore_transitions), _stream(labels)), "tem_affinity_target")
    return out


def standardize(x: torch.Tensor, eps: float = 1e-7) -> torch.Tensor:
    """Per-sample (first axis) standardisation of a contiguous tensor."""
    _req_cuda(x)
    x = x.contiguous()
    N = x.shape[0]
    L = x.numel() // N
    y = torch.empty_like(x)
    lib = _lib.load()
    nws = N * 256 * 16
    ws = _workspace(nws, x.device)
    _lib.check(lib.tem_standardize(_p(x), _p(y), N, L, eps, _p(ws), nws, _stream(x)), "tem_standardize")
    return y
