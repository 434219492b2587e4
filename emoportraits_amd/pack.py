"""Load-time weight preparation: spectral-norm / weight-standardisation folding and packing of conv
weights into the layout the implicit-GEMM kernel stages with 16-byte copies (emoportraits_amd/csrc/conv_igemm.h).

Everything here runs once per checkpoint, on the host, in plain torch fp32 -- it is data preparation, not the hot
path.  The folding formulas restate what the reference recomputes on every forward:
  spectral norm  utils/spectral_norm.py:96-168 (eval: no power iteration)   W / (u . (W_mat v))
  weight std     networks/volumetric_avatar/utils.py:893-900, :908-914      (W - mean) / (std_unbiased + 1e-5)
"""
import contextlib
import ctypes
import functools
import math

import torch

from . import hip

CFG_A, CFG_B, CFG_C, CFG_D, CFG_E, CFG_F, CFG_G = 0, 1, 2, 3, 4, 5, 6   # G: fp16-operand 3x3 kernel only (128 x 256 tile)
_BM = {CFG_A: 128, CFG_B: 64, CFG_C: 32, CFG_D: 64, CFG_E: 64, CFG_F: 32, CFG_G: 128}            # output channels per block
_BP = {CFG_A: 128, CFG_B: 128, CFG_C: 128, CFG_D: 256, CFG_E: 512, CFG_F: 256, CFG_G: 256}       # output positions per block
_PACK_AS = {CFG_D: CFG_B, CFG_E: CFG_B, CFG_F: CFG_C}                                 # configs sharing another one's weight layout


def fold_sn(weight_orig, u, v):
    w_mat = weight_orig.reshape(weight_orig.shape[0], -1)
    sigma = torch.dot(u, torch.mv(w_mat, v))
    return weight_orig / sigma


def fold_ws(w):
    m = w
    for dim in range(1, w.dim()):
        m = m.mean(dim=dim, keepdim=True)
    w = w - m
    std = w.view(w.size(0), -1).std(dim=1).view(-1, *([1] * (w.dim() - 1))) + 1e-5
    return w / std.expand_as(w)


def folded_conv(sd, prefix, kind):
    """(weight, bias) of the conv at `prefix` in a raw reference state_dict; kind in {'sn','ws','plain'}"""
    if kind == "sn":
        w = fold_sn(sd[prefix + ".weight_orig"].float(), sd[prefix + ".weight_u"].float(), sd[prefix + ".weight_v"].float())
    elif kind == "ws":
        w = fold_ws(sd[prefix + ".weight"].float())
    elif kind == "plain":
        w = sd[prefix + ".weight"].float()
    else:
        raise ValueError(kind)
    b = sd.get(prefix + ".bias")
    return w, (None if b is None else b.float())


def pack_generic(w):
    """[Cout,Cin,KH,KW] -> [Cin*KH*KW, CoutP] (CoutP = Cout rounded up to 64, zero padded): the operand layout of
    emo_conv2d_generic_f32 (include/emo_hip.h)"""
    cout = w.shape[0]
    coutp = (cout + 63) // 64 * 64
    wt = torch.zeros((w[0].numel(), coutp), dtype=torch.float32)
    wt[:, :cout] = w.reshape(cout, -1).t()
    return wt.contiguous()


def choose_cfg(cout):
    """block config minimising padded output channels; ties go to the 64-row tile (see _CFG_EFF), then the 128-row one"""
    best = None
    for cfg in (CFG_B, CFG_A, CFG_C):
        bm = _BM[cfg]
        padded = -(-cout // bm) * bm
        if best is None or padded < best[0]:
            best = (padded, cfg)
    return best[1]


def conv_pack_info(kh, kw, cfg):
    lib = hip.load()
    bm, kc = ctypes.c_int(), ctypes.c_int()
    hip.check(lib.emo_conv_pack_info(kh, kw, cfg, ctypes.byref(bm), ctypes.byref(kc)), "emo_conv_pack_info")
    return bm.value, kc.value


def conv_pack_info_f16(kh, kw, cfg):
    lib = hip.load()
    bm, kc = ctypes.c_int(), ctypes.c_int()
    hip.check(lib.emo_conv_pack_info_f16(kh, kw, cfg, ctypes.byref(bm), ctypes.byref(kc)), "emo_conv_pack_info_f16")
    return bm.value, kc.value


def pack_weight_f16(w, cfg):
    """fp16 operand layout of emo_conv_igemm_f16acc32: [co_tile][cin chunk][kd][q][tap][half][BM][8] with
    channel-in-chunk = 16*q + 8*half + 0..7  ->  flat fp16 tensor (round-to-nearest-even of the folded fp32 weight)"""
    if w.dim() == 4:
        w = w.unsqueeze(2)
    cout, cin, kd, kh, kw = w.shape
    bm, kc = conv_pack_info_f16(kh, kw, cfg)
    n_cot = -(-cout // bm)
    n_cc = -(-cin // kc)
    wp = torch.zeros((n_cot * bm, n_cc * kc, kd, kh * kw), dtype=torch.float32)
    wp[:cout, :cin] = w.reshape(cout, cin, kd, kh * kw).float()
    # [cot, BM, cc, q, half, k8, kd, tap] -> [cot, cc, kd, q, tap, half, BM, k8]
    wp = wp.view(n_cot, bm, n_cc, kc // 16, 2, 8, kd, kh * kw).permute(0, 2, 6, 3, 7, 4, 1, 5).contiguous()
    return wp.view(-1).to(torch.float16)


def split_bf16x3(w):
    """fp32 tensor -> (h, m, l) bf16 tensors with h + m + l == w exactly (round-to-nearest-even at every level; the
    residuals w - h and w - h - m are exact in fp32) -- the operand form of emo_conv_igemm_bf16x3"""
    w = w.float()
    h = w.to(torch.bfloat16)
    r1 = w - h.float()
    m = r1.to(torch.bfloat16)
    l = (r1 - m.float()).to(torch.bfloat16)
    return h, m, l


BF16X3_BM, BF16X3_KC = 64, 16    # emo_conv_pack_info_bf16x3 (checked against the library in tests/test_host_logic.py)


def pack_weight_bf16x3(w):
    """operand layout of emo_conv_igemm_bf16x3: [co_tile][cin chunk of 16][kd][kernel row][plane h|m|l][kernel column]
    [half][BM = 64][8], channel in chunk = 8*half + 0..7 -> flat bf16 tensor"""
    if w.dim() == 4:
        w = w.unsqueeze(2)
    cout, cin, kd, kh, kw = w.shape
    if (kh, kw) != (3, 3):
        raise ValueError("3x3 kernels only")
    bm, kc = BF16X3_BM, BF16X3_KC
    n_cot = -(-cout // bm)
    n_cc = -(-cin // kc)
    wp = torch.zeros((n_cot * bm, n_cc * kc, kd, kh, kw), dtype=torch.float32)
    wp[:cout, :cin] = w.float()
    planes = torch.stack(split_bf16x3(wp), 0)                     # [plane, co, ci, kd, r, s]
    # [plane, cot, BM, cc, half, k8, kd, r, s] -> [cot, cc, kd, r, plane, s, half, BM, k8]
    planes = planes.view(3, n_cot, bm, n_cc, 2, 8, kd, kh, kw).permute(1, 3, 6, 7, 0, 8, 4, 2, 5).contiguous()
    return planes.view(-1)


F16X2_GUARD_DEFAULT = __import__("os").environ.get("EMO_F16X2_GUARD", "1") != "0"   # ops.F16X2_GUARD starts from this
F16X2_BM32 = __import__("os").environ.get("EMO_F16X2_BM32", "1") != "0"   # A/B switch: 32-channel layers on the half-empty 64-row tile
F16X2_POINTWISE = __import__("os").environ.get("EMO_F16X2_POINTWISE", "1") != "0"   # A/B switch: 1x1 layers of an f16x2 model on the fp32 MFMA kernel
F16X2_IN_SCALE = 32.0      # emo_conv_igemm_f16x2: the staged input is multiplied by this (inputs beyond +-2047 saturate)

# Overflow flags of the fp16-split layers (include/emo_hip.h, emo_conv_igemm_f16x2): one int32 word per layer in a per-device
# pool.  A launch raises its layer's word when a staged value left the fp16 range; the guarded emo_conv_igemm_bf16x3 launch that
# ops.conv_igemm issues right behind it then recomputes the layer with exact operands.  Words are sticky until
# clear_overflow_flags(): a stale 1 only costs an unnecessary (correct) recomputation.
_FLAG_POOL_WORDS = 4096
_flag_pools = {}     # device -> [int32 tensor, next free slot, {slot: layer name}]


def _flag_pool(device):
    import torch as _t
    dev = _t.device(device)
    if dev.type == "cuda" and dev.index is None:      # 'cuda' and 'cuda:<current>' are one pool (tensors report the indexed form)
        dev = _t.device("cuda", _t.cuda.current_device())
    key = str(dev)
    if key not in _flag_pools:
        _flag_pools[key] = [_t.zeros(_FLAG_POOL_WORDS, dtype=_t.int32, device=device), 0, {}]
    return _flag_pools[key]


def overflow_flag_slot(device, name=None):
    pool = _flag_pool(device)
    slot = pool[1] % _FLAG_POOL_WORDS        # (wraps after 4096 layers: sharing a word is conservative, never wrong)
    pool[1] += 1
    pool[2][slot] = name
    return slot


def overflow_flag_ptr(device, slot):
    return ctypes.c_void_p(_flag_pool(device)[0].data_ptr() + 4 * slot)


def clear_overflow_flags(device):
    """zero every overflow word of the device's pool (stream-ordered: one fill kernel).  HotPath / Stage2 call it at the start
    of a pass."""
    _flag_pool(device)[0].zero_()


def overflow_events(device):
    """{slot: layer name} of the words that are raised right now (host synchronisation: diagnostics and tests)"""
    pool = _flag_pool(device)
    raised = pool[0].nonzero().flatten().tolist()
    return {s: pool[2].get(s) for s in raised}


def pack_weight_f16x2(w, bm=None):
    """operand layout of emo_conv_igemm_f16x2: the bf16x3 layout with two fp16 planes of w * w_scale, w_scale = the power of
    two that puts max|w| into [512, 1024) -> (flat fp16 tensor, w_scale).  bm: channel tile height, 64 (block config D) or 32
    (block config F: layers with at most 32 output channels per tile)"""
    if w.dim() == 4:
        w = w.unsqueeze(2)
    cout, cin, kd, kh, kw = w.shape
    if (kh, kw) != (3, 3):
        raise ValueError("3x3 kernels only")
    bm, kc = (BF16X3_BM if bm is None else bm), BF16X3_KC
    if bm not in (32, 64):
        raise ValueError("channel tiles of 64 or 32 rows")
    n_cot = -(-cout // bm)
    n_cc = -(-cin // kc)
    wmax = float(w.abs().max())
    w_scale = 2.0 ** math.floor(math.log2(1023.0 / wmax)) if wmax > 0 else 1.0
    wp = torch.zeros((n_cot * bm, n_cc * kc, kd, kh, kw), dtype=torch.float32)
    wp[:cout, :cin] = w.float() * w_scale
    w1 = wp.to(torch.float16)
    w2 = (wp - w1.float()).to(torch.float16)
    planes = torch.stack((w1, w2), 0)
    planes = planes.view(2, n_cot, bm, n_cc, 2, 8, kd, kh, kw).permute(1, 3, 6, 7, 0, 8, 4, 2, 5).contiguous()
    return planes.view(-1), w_scale


def pack_weight_f16w8(w):
    """operand layout of emo_conv_igemm_f16w8 (plain fp16 operands on the eight-wave two-tile kernel): the fp16-split layout with
    its two PLANES holding the two 16-channel K BLOCKS of a 32-channel stage,
    [co_tile][cin chunk of 32][kd][kernel row][k-block][kernel column][half][64][8] fp16 of w * w_scale (channel in chunk =
    16 * k-block + 8 * half + 0..7) -> (flat fp16 tensor, w_scale)"""
    if w.dim() == 4:
        w = w.unsqueeze(2)
    cout, cin, kd, kh, kw = w.shape
    if (kh, kw) != (3, 3):
        raise ValueError("3x3 kernels only")
    bm, kc = BF16X3_BM, 2 * BF16X3_KC
    n_cot = -(-cout // bm)
    n_cc = -(-cin // kc)
    wmax = float(w.abs().max())
    w_scale = 2.0 ** math.floor(math.log2(1023.0 / wmax)) if wmax > 0 else 1.0
    wp = torch.zeros((n_cot * bm, n_cc * kc, kd, kh, kw), dtype=torch.float32)
    wp[:cout, :cin] = w.float() * w_scale
    w1 = wp.to(torch.float16)
    # [cot, BM, cc, kblk, half, k8, kd, r, s] -> [cot, cc, kd, r, kblk, s, half, BM, k8]
    w1 = w1.view(n_cot, bm, n_cc, 2, 2, 8, kd, kh, kw).permute(0, 2, 6, 7, 3, 8, 4, 1, 5).contiguous()
    return w1.view(-1), w_scale


F16_W8 = __import__("os").environ.get("EMO_F16_W8", "1") != "0"   # A/B switch: 0 keeps every fp16-operand layer on conv_igemm_f16.h
# odd channel-tile counts on it (the last pair half empty): "5" (default) from five tiles on -- a sixth of the launch's staging is
# wasted, and it wins (320 outputs: +8 / +17 % with the early patch loads, tools/session/r6_call24.sh); "1": every odd count
# (three tiles waste a quarter: 0.9-1.0x the older kernel; one tile half of everything); "0": even counts only
F16_W8_ODD = {"0": 0, "1": 1, "5": 5}.get(__import__("os").environ.get("EMO_F16_W8_ODD", "5"), 5)


# an odd channel-tile count >= 3: the whole pairs on the eight-wave kernel, the last tile on conv_igemm_f16.h
# (emo_conv_igemm_f16w8_rest, ABI 10) -- no half-empty pair; 0: A/B switch (F16_W8_ODD then decides as before)
F16_W8_REST = __import__("os").environ.get("EMO_F16_W8_REST", "1") != "0"


def f16w8_rest_fits(cout, Hl, Wl):
    """emo_conv_igemm_f16w8_rest takes the layer: an odd tile count >= 3 and an output plane that is in the older kernel's
    launch form too (f16_launch_fits: its 2 x 128 / 4 x 64 tiles)"""
    cot = cout // BF16X3_BM
    return F16_W8_REST and cout % BF16X3_BM == 0 and cot >= 3 and cot % 2 == 1 and Hl is not None and f16_launch_fits(Hl, Wl)


def f16w8_launch_fits(cout, cin, kd, kh, kw, Hl, Wl, n_pos_tiles, act="none", positions_per_sample=0):
    """the one launch form of emo_conv_igemm_f16w8 (conv_f16x2_w8_launch<.., NPROD = 1> -- every check of the C launcher has its
    mirror here): 3x3 / 3x3x3, whole 64-channel tiles and 8-channel groups, 4 x 64 position tiles, no activation, at most 2^23
    positions per sample, two pair items per CU.  And -- a choice of the planner, not a limit of the kernel -- an even number of
    channel tiles, or an odd one from F16_W8_ODD tiles on: the kernel runs an odd last tile in a half-empty pair, which costs a
    whole pair's staging"""
    if not F16_W8 or (kh, kw) != (3, 3) or kd not in (1, 3) or cout % BF16X3_BM or cin % 8 or act != "none":
        return False
    if cout % (2 * BF16X3_BM) and not (F16_W8_ODD and cout // BF16X3_BM >= F16_W8_ODD) \
            and not f16w8_rest_fits(cout, Hl, Wl):
        return False
    if Hl is None or Wl % 64 or Hl % 4 or positions_per_sample > (1 << 23):
        return False
    min_items = int(__import__("os").environ.get("EMO_CONV_CT2_MIN_ITEMS", 2 * cu_count()))
    cot = cout // BF16X3_BM
    pairs = cot // 2 if f16w8_rest_fits(cout, Hl, Wl) else -(-cot // 2)
    return (n_pos_tiles // 2) * pairs >= min_items


F16X2_P1_KC = 32     # conv_igemm_f16x2_p1.h: input channels per stage of the pointwise kernel


def pack_weight_f16x2_1x1(w):
    """operand layout of emo_conv_igemm_f16x2 for KH = KW = 1 (csrc/conv_igemm_f16x2_p1.h):
    [channel tile, padded to an EVEN count][Cin chunk of 32][plane 1|2][k-step of 16][half][BM = 64][8] fp16 of w * w_scale
    -> (flat fp16 tensor, w_scale).  One (tile, chunk) block is 8 KB, copied into LDS as it lies."""
    w = w.reshape(w.shape[0], w.shape[1])
    cout, cin = w.shape
    bm, kc = BF16X3_BM, F16X2_P1_KC
    n_cot = -(-cout // bm)
    n_cot += n_cot & 1                          # (an odd last tile runs in a pair whose second half is zero weights)
    n_cc = -(-cin // kc)
    wmax = float(w.abs().max())
    w_scale = 2.0 ** math.floor(math.log2(1023.0 / wmax)) if wmax > 0 else 1.0
    wp = torch.zeros((n_cot * bm, n_cc * kc), dtype=torch.float32)
    wp[:cout, :cin] = w.float() * w_scale
    w1 = wp.to(torch.float16)
    w2 = (wp - w1.float()).to(torch.float16)
    planes = torch.stack((w1, w2), 0)           # [plane, co, ci]
    # [plane, cot, BM, cc, ks, half, k8] -> [cot, cc, plane, ks, half, BM, k8]
    planes = planes.view(2, n_cot, bm, n_cc, 2, 2, 8).permute(1, 3, 0, 4, 5, 2, 6).contiguous()
    return planes.view(-1), w_scale


def pack_weight(w, cfg):
    """w [Cout, Cin, KH, KW] or [Cout, Cin, KD, KH, KW] -> flat fp32 tensor
    [co_tile][cin chunk][kd][pair][tap][half][BM]   (stage index = chunk*KD + kd; k-local = (pair*TAPS+tap)*2+half)"""
    if w.dim() == 4:
        w = w.unsqueeze(2)
    cout, cin, kd, kh, kw = w.shape
    bm, kc = conv_pack_info(kh, kw, cfg)
    n_cot = -(-cout // bm)
    n_cc = -(-cin // kc)
    wp = torch.zeros((n_cot * bm, n_cc * kc, kd, kh * kw), dtype=torch.float32)
    wp[:cout, :cin] = w.reshape(cout, cin, kd, kh * kw).float()
    # [cot, BM, cc, pair, half, kd, tap] -> [cot, cc, kd, pair, tap, half, BM]
    wp = wp.view(n_cot, bm, n_cc, kc // 2, 2, kd, kh * kw).permute(0, 2, 5, 3, 6, 4, 1).contiguous()
    return wp.view(-1)


# relative MFMA efficiency of the block configs measured on MI355X at 16 frames (archive/profiles/r2_conv_microbench.jsonl):
# the 64-row tile (15 KB stage buffers -> 5 blocks per CU) beats the 128-row one (25 KB -> 3 blocks) on the 64^2 .. 256^2
# layers (133 vs 124 TF on 512->512 @64^2) and ties it at 512^2 (132-135); the 32-row tile re-stages the same input patch
# for a quarter of the work
# the 64 x 256 tile (D) halves the weight-tile traffic per MFMA and stages 25-33 % less patch per position: 131-141 TF on
# every 2-D 3x3 decoder layer (B: 126-135), bench 127.3 -> 132.0 frames/s (archive/profiles/r2_conv_microbench.jsonl)
_CFG_EFF = {CFG_A: 0.97, CFG_B: 1.0, CFG_C: 0.88, CFG_D: 1.03, CFG_E: 0.0, CFG_F: 0.92, CFG_G: 0.0}
if __import__("os").environ.get("EMO_F16_CFG_G") == "1":   # opt-in: plan fp16-operand 3x3 layers with the 128 x 256 tile
    _CFG_EFF[CFG_G] = 1.15                                  # (one block per CU; built and parity-tested, not yet the default)
if __import__("os").environ.get("EMO_CONV_CFG_D") == "0":   # A/B switch: plan without the 64 x 256 tile
    _CFG_EFF[CFG_D] = 0.0
if __import__("os").environ.get("EMO_CONV_CFG_E") == "1":   # A/B switch: plan with the 64 x 512 tile
    _CFG_EFF[CFG_E] = 1.06


@functools.lru_cache(maxsize=None)
def cu_count():
    """compute units the C launchers size their grids by (emo_device_cu_count, ABI 9: the current device's, a multiple of 8; 256
    on MI355X, 32 .. 128 on its partitions; 256 without a device) -- the planner's fill targets follow it, so that Python and C
    agree on when the pointwise / two-tile launch forms pay off.  Read once per process (one process per GPU)."""
    try:
        return int(hip.load().emo_device_cu_count())
    except Exception:
        return 256


def _fill_blocks():
    return 2 * cu_count()   # 2 blocks per CU


def cfg_d_fits(kd, kh, kw, Hl, Wl):
    """the 64 x 256 (and 32 x 256) tiles exist for 3x3 / 3x3x3 fp32 layers whose output planes are tiled by 2x128 / 4x64 /
    8x32 positions (the depth taps of a 3-D layer run as K stages, one depth slice per tile)"""
    if kd not in (1, 3) or (kh, kw) != (3, 3) or Hl is None:
        return False
    return (Wl % 128 == 0 and Hl % 2 == 0) or (Wl == 64 and Hl % 4 == 0) or (Wl == 32 and Hl % 8 == 0)


def cfg_e_fits(kd, kh, kw, Hl, Wl):
    if kd != 1 or (kh, kw) != (3, 3) or Hl is None:
        return False
    return (Wl % 128 == 0 and Hl % 4 == 0) or (Wl == 64 and Hl % 8 == 0)


def choose_cfg_for_launch(cout, n_pos_tiles, allowed=(CFG_A, CFG_B, CFG_C)):
    """pick the block config for one launch: enough blocks to fill 256 CUs first, then least channel padding,
    then the larger tile"""
    best = None
    for cfg in allowed:
        bm = _BM[cfg]
        cot = -(-cout // bm)
        blocks = cot * n_pos_tiles
        score = min(blocks, _fill_blocks()) / _fill_blocks() * (cout / (cot * bm)) * _CFG_EFF[cfg]
        if best is None or score > best[0] + 1e-9:
            best = (score, cfg)
    return best[1]


@functools.lru_cache(maxsize=None)
def _kc(kh, kw, cfg, precision="f32"):
    """input channels per K stage of a block config (emo_conv_pack_info / emo_conv_pack_info_f16)"""
    if precision in ("bf16x3", "f16x2"):
        return BF16X3_KC
    return (conv_pack_info_f16 if precision == "f16" else conv_pack_info)(kh, kw, cfg)[1]


_MAX_KSPLIT = 16


def ksplit_for(blocks, nstages):
    """split the K loop until the launch has two blocks per CU, keeping >= 8 stages per split
    (same rule as emo_conv_igemm_ksplit in csrc/conv_api.hip)"""
    fill = _fill_blocks()
    if blocks >= fill:
        return 1
    return max(1, min(-(-fill // blocks), nstages // 8, _MAX_KSPLIT))


def _quantisation(nblocks, cus=None):
    """a launch of a few blocks per CU finishes when the most loaded CU does: 640 blocks put 3 on half of the CUs and 2 on
    the rest -- 2.5 / 3 of the machine.  Measured at batch 2 (archive/profiles/r3_conv_microbench_b2.jsonl): 320 -> 320 @128^2 runs at
    113 TF on the 64 x 256 tile (640 blocks) and at 130 TF on the 64 x 128 tile (1280 blocks)."""
    if cus is None:
        cus = cu_count()
    if nblocks < cus or EMO_PLAN_QUANTISATION == 0:
        return 1.0
    per_cu = nblocks / cus
    return per_cu / math.ceil(per_cu - 1e-9)


EMO_PLAN_QUANTISATION = int(__import__("os").environ.get("EMO_PLAN_QUANTISATION", "1"))   # 0: A/B switch (round-2 planner)


def plan_launch(cout, cin, kd, kh, kw, n_pos_tiles, allowed=(CFG_A, CFG_B, CFG_C), precision="f32"):
    """(block config, K split) of one launch: fill the 256 CUs first (by splitting K if the tile grid is small), then
    least channel padding, then the larger tile"""
    best = None
    for cfg in allowed:
        nstages = -(-cin // _kc(kh, kw, cfg, precision)) * kd
        bm = _BM[cfg]
        cot = -(-cout // bm)
        blocks = cot * max(1, n_pos_tiles * 128 // _BP[cfg])
        ks = ksplit_for(blocks, nstages)
        score = min(blocks * ks, _fill_blocks()) / _fill_blocks() * (cout / (cot * bm)) * _CFG_EFF[cfg] * (0.97 if ks > 1 else 1.0)
        score *= _quantisation(blocks * ks)
        if best is None or score > best[0] + 1e-9:
            best = (score, cfg, ks)
    return best[1], best[2]


_build_precision = "f32"
# 'f32': the exact-fp32 MFMA kernel everywhere.  'bf16x3': fp32 results on the bf16 matrix pipes -- operands split exactly into
# three bf16 terms, six partial products, fp32 accumulation (csrc/conv_igemm_bf16x3.h) -- on the 3x3 layers that kernel covers,
# the exact-fp32 kernel elsewhere.  'f16': reduced precision (fp16 operands), opt-in.
# 'f16x2': companion of 'bf16x3' with half the matrix work -- the scaled operands as two fp16 terms (2^-24 relative), three
# products.  Its operand range (inputs beyond +-2047 after norm + ReLU saturate) is checked ON THE DEVICE by every launch, and a
# guarded bf16x3 launch of the same layer recomputes it when the check fires (include/emo_hip.h, emo_conv_igemm_f16x2).
PRECISIONS = ("f32", "f16", "bf16x3", "f16x2")


@contextlib.contextmanager
def conv_precision(precision):
    """precision requested for the PackedConvs built inside the block ('f32' | 'f16'); layers the fp16-operand kernel
    does not cover (7x7 stem, heads with fewer than 32 output channels) stay fp32.  Used by HotPath / Stage2
    constructors; construction is single-threaded."""
    global _build_precision
    if precision not in PRECISIONS:
        raise ValueError("precision must be one of %s" % (PRECISIONS,))
    old, _build_precision = _build_precision, precision
    try:
        yield
    finally:
        _build_precision = old


def supports_f16(cout, cin, kd, kh, kw):
    """layers the fp16-operand kernel can take at all (whether a given LAUNCH runs on it also depends on the output width:
    PackedConv.plan_for)"""
    return (kh, kw) in ((3, 3), (1, 1)) and kd in (1, 3) and not (kd == 3 and kh == 1) and cout >= 32 and cin % 8 == 0


def supports_bf16x3(cout, cin, kd, kh, kw, precision="bf16x3"):
    """layers the split-operand kernel takes: 3x3 / 3x3x3, whole 8-channel groups, and a 64-row channel tile that is at least
    three quarters real in the bf16 split (below that the exact-fp32 kernel's 32-row tiles win: six products on a half-empty
    tile are twelve per useful one), at least half real in the fp16 split (three products: the 32-channel 3-D layers of the
    WarpGenerator run 1.6x the fp32 MFMA kernel's speed on a half-empty tile)"""
    fill = cout / (-(-cout // BF16X3_BM) * BF16X3_BM)
    if precision == "f16x2" and F16X2_BM32 and cout <= 32:      # (a 32-row tile; also the 3-channel warp head: a tenth of the tile is
        fill = 1.0                                              # real, and it is still 3x the fp32 MFMA kernel's 32-row tile)
    return (kh, kw) == (3, 3) and kd in (1, 3) and cin % 8 == 0 and fill >= (0.5 if precision == "f16x2" else 0.75)


def f16x2_tile_cfg(cout):
    """block config of an fp16-split 3x3 layer: F (32 channels x 256 positions, csrc/conv_igemm_bf16x3.h BMT = 32) for layers
    with at most 32 output channels -- they ran the 64-row tile half empty --, D (64 x 256) otherwise"""
    return CFG_F if (F16X2_BM32 and cout <= 32) else CFG_D


def supports_f16x2_pointwise(cout, cin, kd, kh, kw):
    """pointwise layers the fp16 split takes (csrc/conv_igemm_f16x2_p1.h): whole 64-channel tiles, at least one PAIR of them,
    whole 8-channel input groups, an even number of 32-channel stages"""
    return (kd, kh, kw) == (1, 1, 1) and cout % BF16X3_BM == 0 and cout >= 2 * BF16X3_BM and cin % 8 == 0 \
        and (-(-cin // F16X2_P1_KC)) % 2 == 0          # (its K loop runs two 32-channel stages per iteration)


def f16x2_pointwise_launch_fits(Hl, Wl, ups, n_pos_tiles, cout, act="none", positions_per_sample=0):
    """the launch form of the pointwise kernel (conv_f16x2_p1_launch, csrc/conv_igemm_f16x2_p1.h -- every check of the C launcher
    has its mirror here, so that a launch it would refuse with EMO_ERR_UNSUPPORTED is planned onto the fp32 MFMA kernel instead):
    4 x 64 position tiles on the source grid, no activation, at most 2^23 positions per sample (its 32-bit offsets), and enough
    pair items for two per CU of THIS device (below that the fp32 MFMA kernel's K split fills the chip better)"""
    if Hl is None or ups or act != "none" or Wl % 64 or Hl % 4 or positions_per_sample > (1 << 23):
        return False
    min_items = int(__import__("os").environ.get("EMO_F16X2_P1_MIN_ITEMS", 2 * cu_count()))      # (tests lower it to reach small shapes)
    return (n_pos_tiles // 2) * (-(-(cout // BF16X3_BM) // 2)) >= min_items


def bf16x3_launch_fits(Hl, Wl, ups=False):
    """output planes tiled by 4 x 64 positions, or (32- / 16-wide maps, no fused upsample) 8 x 32 / 16 x 16"""
    return Hl is not None and ((Wl % 64 == 0 and Hl % 4 == 0) or (Wl == 32 and Hl % 8 == 0 and not ups)
                               or (Wl == 16 and Hl % 16 == 0 and not ups))


F16_AFFINE_MAX_CIN = 1024   # ConvCfgH::SCT (conv_igemm_f16.h)


def f16_launch_fits(Hl, Wl):
    """output planes tiled by the 64 x 256 tile of the fp16-operand kernel: 2x128 / 4x64 / 8x32 positions"""
    if Hl is None:
        return False
    return (Wl % 128 == 0 and Hl % 2 == 0) or (Wl == 64 and Hl % 4 == 0) or (Wl == 32 and Hl % 8 == 0)


class PackedConv:
    """One convolution of the hot path, ready for emo_conv_igemm_f32.  Weights are packed lazily per block config
    (the best config depends on the batch size of the call); `cfg` pins one config (tests / benchmarks).
    precision="f16" (opt-in, BASELINE configs[4]): fp16 MFMA operands with fp32 accumulation, emo_conv_igemm_f16acc32, on
    every launch whose output planes the 64 x 256 tile covers (widths that are multiples of 128, or 64 / 32); other launches
    of the layer (the 16- and 8-wide WarpGenerator maps) run the exact-fp32 kernel."""

    def __init__(self, name, weight, bias, device, cfg=None, precision="f32"):
        if weight.dim() == 4:
            cout, cin, kh, kw = weight.shape
            kd = 1
        else:
            cout, cin, kd, kh, kw = weight.shape
        self.name = name
        self.cin, self.cout, self.kd, self.kh, self.kw = cin, cout, kd, kh, kw
        self.device = device
        self._weight = weight.float().contiguous()      # folded fp32 weight kept on the host for lazy packing
        self._packed = {}
        self.pinned_cfg = cfg
        if precision not in PRECISIONS:
            raise ValueError("precision must be one of %s" % (PRECISIONS,))
        self.pointwise_split = precision == "f16x2" and F16X2_POINTWISE and supports_f16x2_pointwise(cout, cin, kd, kh, kw)
        if precision == "f16x2" and not F16X2_POINTWISE and (kd, kh, kw) == (1, 1, 1):
            precision = "f32"           # (EMO_F16X2_POINTWISE=0: the A/B switch also holds for an explicitly requested precision)
        if precision in ("bf16x3", "f16x2") and not self.pointwise_split and not supports_bf16x3(cout, cin, kd, kh, kw, precision):
            raise ValueError(f"{name}: the split-operand kernel covers 3x3 / 3x3x3 convolutions with a multiple of 8 input channels "
                             f"(fp16 split: also 1x1 layers with at least 128 output channels)")
        if precision == "f16" and not supports_f16(cout, cin, kd, kh, kw):
            raise ValueError(f"{name}: the fp16-operand kernel covers 3x3 / 3x3x3 / 1x1 convolutions with >= 32 output "
                             f"channels and a multiple of 8 input channels")
        self.precision = precision
        self.allowed = (CFG_A, CFG_B) if (kh, kw) == (1, 7) else (CFG_A, CFG_B, CFG_C)
        self.bias = None if bias is None else bias.float().contiguous().to(device)
        self.macs_per_position = cout * cin * kd * kh * kw
        if precision == "f32":
            first = choose_cfg(cout) if cfg is None else cfg
            self.packed(first if first in self.allowed else CFG_B)
        elif self.pointwise_split:
            # pointwise layer on the fp16 split: its own operand layout; the fp32 layout as well -- launches the pointwise kernel
            # does not take (small batches, activations), and the guarded exact recomputation, run the fp32 MFMA kernel
            self.packed(CFG_D, "f16x2")
            first = choose_cfg(cout) if cfg in (None, CFG_D) else cfg
            self.packed(first if first in self.allowed else CFG_B)
            self.flag_slot = overflow_flag_slot(device, name)
        elif precision in ("bf16x3", "f16x2"):
            # eager, like the fp32 layout: the first launch is not a host-side packing job
            self.packed(f16x2_tile_cfg(cout) if precision == "f16x2" else CFG_D, precision)
            if precision == "f16x2":            # the guarded exact recomputation behind a raised overflow flag (ops.conv_igemm)
                if F16X2_GUARD_DEFAULT:         # (EMO_F16X2_GUARD=0 builds never launch it: packed lazily if a caller turns
                    self.packed(CFG_D, "bf16x3")    # the guard on later -- 1.5x the fp16 planes' memory otherwise)
                self.flag_slot = overflow_flag_slot(device, name)

    def plain_weight(self):
        """the folded weight as [Cout, Cin * KD * KH * KW] fp32 on the device (ops.conv_head: no packed layout)"""
        if "plain" not in self._packed:
            self._packed["plain"] = self._weight.reshape(self.cout, -1).contiguous().to(self.device)
        return self._packed["plain"]

    def packed(self, cfg, precision="f32"):
        """packed weights for a block config: fp32 layout, or the fp16 operand layout (64 x 256 and 128 x 256 tiles)"""
        if precision == "f16w8":
            if "f16w8" not in self._packed:
                flat, self.w_scale16 = pack_weight_f16w8(self._weight)
                self._packed["f16w8"] = flat.to(self.device)
            return self._packed["f16w8"]
        if precision == "f16":
            key = ("f16", CFG_G if cfg == CFG_G else CFG_D)
            if key not in self._packed:
                self._packed[key] = pack_weight_f16(self._weight, key[1]).to(self.device)
            return self._packed[key]
        if precision == "bf16x3":
            if "bf16x3" not in self._packed:
                self._packed["bf16x3"] = pack_weight_bf16x3(self._weight).to(self.device)
            return self._packed["bf16x3"]
        if precision == "f16x2":
            # keyed by the channel tile height: a layer with <= 32 output channels is packed for the 32-row tile (block config F)
            # eagerly and -- lazily, like the guard's bf16x3 weights -- for the 64-row tile (D) that its launches on 32- / 16-wide
            # maps or with a fused upsample run (round 5 sent those to the fp32 MFMA kernel: another speed AND another
            # arithmetic path).  w_scale depends on max|w| only: the same for both layouts
            bm = BF16X3_BM if (self.pointwise_split or cfg != CFG_F) else 32
            key = ("f16x2", bm)
            if key not in self._packed:
                if self.pointwise_split:
                    flat, self.w_scale = pack_weight_f16x2_1x1(self._weight)
                else:
                    flat, self.w_scale = pack_weight_f16x2(self._weight, bm=bm)
                self._packed[key] = flat.to(self.device)
            return self._packed[key]
        cfg = _PACK_AS.get(cfg, cfg)
        if cfg not in self._packed:
            self._packed[cfg] = pack_weight(self._weight, cfg).to(self.device)
        return self._packed[cfg]

    def cfg_for(self, n_pos_tiles):
        if self.pinned_cfg is not None:
            return self.pinned_cfg
        return choose_cfg_for_launch(self.cout, n_pos_tiles, self.allowed)

    def plan_for(self, n_pos_tiles, Hl=None, Wl=None, ups=False, affine=False, aligned16=True, in_elems_per_sample=0, act="none",
                 io_aligned16=True):
        """(cfg, ksplit, precision) for a launch over n_pos_tiles 128-position tiles of an Hl x Wl output; `affine`: the
        launch carries a per-sample input scale / shift (the fp16-operand kernel keeps those in a 1024-entry LDS table);
        `aligned16`: the input pointer is 16-byte aligned (that kernel loads 16-byte quads); `in_elems_per_sample`:
        Cin*D*H*W of the input -- that kernel addresses a sample with 32-bit byte offsets (conv_igemm_f16_launch), larger
        inputs take the exact-fp32 kernel like every other unsupported case; `io_aligned16`: output and residual pointers are
        16-byte aligned too (the straight-line epilogue of emo_conv_igemm_f16w8 needs it)"""
        if self.precision == "f16" and self.pinned_cfg in (None, CFG_D) and aligned16 and io_aligned16 \
                and in_elems_per_sample * 4 < (1 << 32) and not (affine and self.cin > F16_AFFINE_MAX_CIN) \
                and f16w8_launch_fits(self.cout, self.cin, self.kd, self.kh, self.kw, Hl, Wl, n_pos_tiles, act,
                                      in_elems_per_sample // max(1, self.cin) * (4 if ups else 1)):
            # the decoders' 3x3 layers in their launch form: plain fp16 operands on the eight-wave two-tile kernel (round 6)
            return CFG_D, 1, "f16w8"
        if self.precision == "f16" and f16_launch_fits(Hl, Wl) and self.pinned_cfg in (None, CFG_D, CFG_G) and aligned16 \
                and in_elems_per_sample * 4 < (1 << 32) \
                and not (affine and self.cin > F16_AFFINE_MAX_CIN) and not (self.pinned_cfg == CFG_G and self.kh != 3):
            tiles = (self.pinned_cfg,) if self.pinned_cfg is not None else \
                (CFG_D, CFG_G) if (_CFG_EFF[CFG_G] > 0 and self.kh == 3) else (CFG_D,)
            cfg, ks = plan_launch(self.cout, self.cin, self.kd, self.kh, self.kw, n_pos_tiles, tiles, "f16")
            return cfg, ks, "f16"
        if self.pointwise_split:
            if f16x2_pointwise_launch_fits(Hl, Wl, ups, n_pos_tiles, self.cout, act, in_elems_per_sample // max(1, self.cin)) \
                    and self.pinned_cfg in (None, CFG_D) \
                    and aligned16 and in_elems_per_sample * 4 < (1 << 32) and not (affine and self.cin > F16_AFFINE_MAX_CIN):
                return CFG_D, 1, "f16x2"
        elif self.precision in ("bf16x3", "f16x2") and bf16x3_launch_fits(Hl, Wl, ups) and self.pinned_cfg in (None, CFG_D) \
                and aligned16 and in_elems_per_sample * 4 < (1 << 32) and not (affine and self.cin > F16_AFFINE_MAX_CIN):
            tile = CFG_D
            if self.precision == "f16x2" and f16x2_tile_cfg(self.cout) == CFG_F and Wl % 64 == 0 and not ups:
                # (the 32-row tile exists for 4 x 64 position tiles without fused upsample; other maps of such a layer run the
                # half-empty 64-row tile -- still the fp16 split's arithmetic, 1.3-1.6x the fp32 MFMA kernel -- from a 64-row
                # packing made on first use: PackedConv.packed)
                tile = CFG_F
            cfg, ks = plan_launch(self.cout, self.cin, self.kd, self.kh, self.kw, n_pos_tiles, (tile,), self.precision)
            return cfg, ks, self.precision
        pinned = None if self.pinned_cfg == CFG_G else self.pinned_cfg   # (G exists for fp16 operands only)
        allowed = (pinned,) if pinned is not None else self.allowed
        if self.pinned_cfg is None and _CFG_EFF[CFG_D] > 0 and cfg_d_fits(self.kd, self.kh, self.kw, Hl, Wl):
            allowed = allowed + (CFG_D,)
            if _CFG_EFF[CFG_F] > 0 and (not ups or Wl % 128 == 0):
                allowed = allowed + (CFG_F,)
        if self.pinned_cfg is None and _CFG_EFF[CFG_E] > 0 and cfg_e_fits(self.kd, self.kh, self.kw, Hl, Wl):
            allowed = allowed + (CFG_E,)
        cfg, ks = plan_launch(self.cout, self.cin, self.kd, self.kh, self.kw, n_pos_tiles, allowed, "f32")
        return cfg, ks, "f32"

    @classmethod
    def from_state_dict(cls, sd, prefix, kind, device, cfg=None, precision=None):
        """precision None: the ambient conv_precision() request, applied where the fp16-operand kernel exists"""
        w, b = folded_conv(sd, prefix, kind)
        if precision is None:
            kd = w.shape[2] if w.dim() == 5 else 1
            split = _build_precision in ("bf16x3", "f16x2")
            ok = supports_bf16x3(w.shape[0], w.shape[1], kd, w.shape[-2], w.shape[-1], _build_precision) if split \
                else supports_f16(w.shape[0], w.shape[1], kd, w.shape[-2], w.shape[-1])
            if _build_precision == "f16x2" and F16X2_POINTWISE and supports_f16x2_pointwise(w.shape[0], w.shape[1], kd, w.shape[-2], w.shape[-1]):
                ok = True
            precision = _build_precision if ok else "f32"
        return cls(prefix, w, b, device, cfg, precision)
