"""sha256 (first 16 hex digits) of the convolution kernel headers: recorded in profiles/r*_pmc_conv_*traffic.json by the script that
takes the counters, compared by bench.py before it quotes such a file as `roofline.traffic` (a PMC figure from a separate run is
evidence for the kernels it was taken on, not for later ones)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ("conv_igemm.h", "conv_igemm_f16.h", "conv_igemm_bf16x3.h", "conv_split_pair_common.h", "conv_igemm_f16x2_ct2.h", "conv_igemm_f16x2_w8.h", "conv_igemm_f16x2_p1.h",
         "conv_api.hip")


def kernel_source_hash():
    h = hashlib.sha256()
    for f in FILES:
        with open(os.path.join(ROOT, "emoportraits_amd", "csrc", f), "rb") as fh:
            h.update(f.encode())
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_hash())
