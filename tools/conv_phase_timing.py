"""Where a block of the split convolution (csrc/conv_igemm_bf16x3.h) spends its time: prologue / K loop / epilogue / store
drain / gap to the next block on the same CU, from s_memtime stamps of wave 0 of every work item.

Needs the measurement build of the library (never the product):
    python -m emoportraits_amd.build --variant timing EMO_S_TIMING=1
    EMO_HIP_LIB=emoportraits_amd/lib/libemoportraits_hip_timing.so python tools/conv_phase_timing.py [B]
JSON lines: per (layer shape, operand mode) the median / p90 of every phase in shader cycles, the per-CU occupancy of each
phase (sum of the phase over the CU's blocks / the CU's busy span), and the launch's wall time by HIP events.
"""
import ctypes
import json
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emoportraits_amd import hip, ops, pack  # noqa: E402

DEV = "cuda:0"


def stamps(lib, mode, n_items):
    fn = getattr(lib, "emo_debug_conv_timing_" + mode)
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    fn.restype = ctypes.c_int
    buf = np.zeros((n_items, 16), dtype=np.uint64)
    hip.check(fn(buf.ctypes.data_as(ctypes.c_void_p), n_items), "emo_debug_conv_timing_" + mode)
    return buf


def analyse_waves(full, n_items):
    """EMO_S_TIMING=2 builds: per wave of a block, the cycles its K loop spent in the waitcnt and in the s_barrier of the
    loop's barriers (rows N/4 + 4 * item + wave of the log)"""
    q4 = full.shape[0] // 4
    n = min(n_items, full.shape[0] // 8)
    ws = full[q4:q4 + 4 * n, :10].astype(np.int64).reshape(n, 4, 10)
    if ws[:, :, 9].any() and (ws[:, :, 9] < 4096).all():      # EMO_S_TIMING=3: per-step cycles (column 9 = stages of the item)
        per_stage = ws[:, :, :9] / ws[:, :, 9:10].clip(min=1)
        return {f"wave{k}": dict(step_cycles=[int(np.median(per_stage[:, k, g])) for g in range(9)]) for k in range(4)}
    w = ws[:, :, :4]
    if not w[:, :, 2].any():
        return None
    out = {}
    for wave in range(4):
        out[f"wave{wave}"] = dict(waitcnt=int(np.median(w[:, wave, 0])), barrier=int(np.median(w[:, wave, 1])),
                                  n_barriers=int(np.median(w[:, wave, 2])), kloop=int(np.median(w[:, wave, 3])))
    return out


def analyse(buf):
    t = buf[:, :12].astype(np.int64)
    hw, xcc = buf[:, 12].astype(np.int64), buf[:, 13].astype(np.int64) & 0xf
    cu = (xcc << 16) | (hw & 0xff00)       # XCC, SE, SH, CU (HW_ID bits 8..15)
    pro, kloop, epi, drain = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3]
    gaps, busy = [], {}
    for c in np.unique(cu):
        idx = np.nonzero(cu == c)[0]
        o = idx[np.argsort(t[idx, 0])]
        if len(o) > 1:
            gaps.append(t[o[1:], 0] - t[o[:-1], 4])
        busy[c] = (t[o[-1], 4] - t[o[0], 0], len(o))
    gaps = np.concatenate(gaps) if gaps else np.zeros(1, dtype=np.int64)
    span = np.array([b[0] for b in busy.values()], dtype=np.float64)
    nblk = np.array([b[1] for b in busy.values()], dtype=np.float64)

    def q(a):
        return dict(med=int(np.median(a)), p10=int(np.percentile(a, 10)), p90=int(np.percentile(a, 90)), mean=round(float(a.mean()), 1))

    tot = float(pro.sum() + kloop.sum() + epi.sum() + drain.sum() + gaps.clip(min=0).sum())
    return dict(n_items=int(len(t)), n_cus=int(len(busy)), blocks_per_cu=round(float(nblk.mean()), 2),
                prologue=q(pro), kloop=q(kloop), epilogue_issue=q(epi), store_drain=q(drain), gap_to_next_block=q(gaps),
                # inside the epilogue: drain of the dead re-issued loads, barrier, residual loads issued, first / second 32
                # channels transposed + stored
                epi_wait_vmcnt0=q(t[:, 5] - t[:, 2]), epi_barrier=q(t[:, 6] - t[:, 5]), epi_res_issue=q(t[:, 7] - t[:, 6]),
                epi_half0=q(t[:, 8] - t[:, 7]), epi_half1=q(t[:, 9] - t[:, 8]), epi_tail=q(t[:, 3] - t[:, 9]),
                share=dict(prologue=round(pro.sum() / tot, 4), kloop=round(kloop.sum() / tot, 4), epilogue_issue=round(epi.sum() / tot, 4),
                           store_drain=round(drain.sum() / tot, 4), gap=round(float(gaps.clip(min=0).sum()) / tot, 4)),
                cu_span_cycles_med=int(np.median(span)))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    lib = hip.load()
    shapes = [(128, 128, (512, 512), False), (192, 128, (256, 256), True), (192, 192, (256, 256), False),
              (320, 320, (128, 128), False), (512, 512, (64, 64), False)]
    if "--shapes" in sys.argv:                                   # e.g. --shapes 0,4
        shapes = [shapes[int(i)] for i in sys.argv[sys.argv.index("--shapes") + 1].split(",")]
    for cin, cout, dims, ups in shapes:
        x = torch.randn(B, cin, *dims, device=DEV)
        w = torch.randn(cout, cin, 3, 3) / math.sqrt(cin * 9)
        if "--zeros" in sys.argv:                                # all-zero operands: the schedule without the power limit
            x.zero_()
            w.zero_()
        scale = torch.rand(B, cin, device=DEV) + 0.5
        shift = torch.randn(B, cin, device=DEV) * 0.1
        odims = tuple(d * 2 for d in dims) if ups else dims
        flops = 2.0 * B * cout * cin * 9 * math.prod(odims)
        n_tiles = B * (math.prod(odims) // 256) * (-(-cout // 64))
        # ct2: the two-tile kernel (conv_igemm_f16x2_ct2.h) on the layer's channel-tile pairs -- one log row per PAIR item, its
        # epilogue stamps 8 / 9 are the first / second tile's; f16x2: the single-tile kernel alone (EMO_CONV_CT2=0)
        # w8: the two-tile kernel with two waves per SIMD (conv_igemm_f16x2_w8.h); its stamps are those of ct2 (7: first tile's
        # residual loads issued, 8 / 9: first / second tile written)
        # f16w8: the same kernel with plain fp16 operands (NPROD = 1, 32-channel stages: precision 'f16' in the decoders' launch form)
        modes = (("bf16x3", "bf16x3"), ("f16x2", "f16x2"), ("ct2", "f16x2"), ("w8", "f16x2"), ("f16w8", "f16"))
        if "--modes" in sys.argv:
            want = sys.argv[sys.argv.index("--modes") + 1].split(",")
            modes = tuple(m for m in modes if m[0] in want)
        for mode, prec in modes:
            if mode in ("ct2", "w8", "f16w8") and ((cout // 64) < 2 or (mode == "f16w8" and (cout // 64) % 2)):
                continue
            os.environ["EMO_CONV_CT2"] = "1" if mode in ("ct2", "w8") else "0"
            os.environ["EMO_CONV_W8"] = "1" if mode == "w8" else "0"
            n_items = B * (math.prod(odims) // 256) * (cout // 128) if mode in ("ct2", "w8", "f16w8") else n_tiles
            layer = pack.PackedConv("t", w, None, DEV, precision=prec)
            gn = None
            real = "--real" in sys.argv          # as the decoder launches it: residual + GroupNorm tile statistics
            with_res = real or "--res" in sys.argv          # (--res / --stats: one of the two alone, to tell their costs apart)
            with_stats = real or "--stats" in sys.argv
            res = torch.randn(B, cout, *odims, device=DEV) if with_res else None
            kw = dict(relu_in=True, ups=ups, res=res, want_stats=with_stats)
            out = ops.conv_igemm(x, layer, scale, shift, **kw)
            out = out[0] if with_stats else out
            for _ in range(2):
                ops.conv_igemm(x, layer, scale, shift, out=out, **kw)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.conv_igemm(x, layer, scale, shift, out=out, **kw)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b)
            rec = dict(B=B, cin=cin, cout=cout, dims=dims, ups=ups, mode=mode, real=real, residual=with_res, statistics=with_stats, stagger=os.environ.get("EMO_CONV_STAGGER", "0"),
                       ms=round(ms, 3), tflops=round(flops / ms / 1e9, 1), tiles_per_item=2 if mode in ("ct2", "w8", "f16w8") else 1,
                       stages=(-(-cin // (32 if mode == "f16w8" else 16))))
            full = stamps(lib, "w8" if mode == "f16w8" else mode, 65536)
            rec.update(analyse(full[:min(n_items, 65536)]))
            waves = analyse_waves(full, n_items)
            if waves is not None:
                rec["waves"] = waves
            rec["eff_clock_ghz"] = round(rec["cu_span_cycles_med"] / (ms * 1e6), 3)
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
