"""Throughput of the stage-2 refinement on the HIP kernels: frames/s at 512x512 for both norm
variants, exact fp32 and the opt-in fp16-operand mode (BASELINE configs[4])."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emoportraits_amd import stage2  # noqa: E402

DEV = "cuda:0"


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    only = sys.argv[2] if len(sys.argv) > 2 else None         # "f16" / "f32": that operand mode alone (kernel traces of one mode)
    for variant, precision in (("bn", "f32"), ("gn_ws", "f32"), ("bn", "f16"), ("gn_ws", "f16")):
        if only is not None and precision != only:
            continue
        over = dict(output_size_s2=512)
        if variant == "gn_ws":
            over.update(norm_layer_type="gn", use_ws=True)
        cfg = stage2.stage2_config(overrides=over)
        s2 = stage2.Stage2(stage2.random_state_dict(cfg, seed=0), cfg, DEV, precision=precision)
        g = torch.Generator().manual_seed(1)
        img = torch.rand(B, 3, 512, 512, generator=g).to(DEV)
        mask = (torch.rand(B, 1, 512, 512, generator=g) > 0.1).float().to(DEV)
        face = (torch.rand(B, 1, 512, 512, generator=g) > 0.3).float().to(DEV)
        for _ in range(2):
            s2.refine(img, mask, face)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            s2.refine(img, mask, face)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 3
        print(json.dumps(dict(stage=2, variant=variant, conv_operands=precision, B=B, ms_per_batch=round(ms, 2), fps=round(B / ms * 1e3, 1))), flush=True)


if __name__ == "__main__":
    main()
