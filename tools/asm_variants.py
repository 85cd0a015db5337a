"""Builds A/B variants of the hand-scheduled kernel: one library per LA_ASM_OPT setting, under build_variants/.

    python tools/asm_variants.py base= nosm=nosoftmax nodma=nodma ...
    (GPU box)  for f in build_variants/*.so; do LA_FWD_KERNEL=asm LITEATTENTION_AMD_LIB=$f python tools/abl_bench.py; done
Ablation variants compute wrong results; they only price a component.
"""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "liteattention_amd", "csrc")
OUT = os.path.join(ROOT, "build_variants")
SRC = ["la_fwd_kernel.hip", "la_fwd_kernel_v2.hip", "la_fwd_kernel_asm.hip", "la_fwd_kernel_w8.hip", "la_fwd_kernel_fp8.hip",
       "la_aux_kernels.hip", "la_api.hip"]


def build_one(spec):
    name, _, opt = spec.partition("=")
    inc = os.path.join(OUT, f"{name}.inc")
    subprocess.run([sys.executable, os.path.join(CSRC, "gen_fwd_asm.py"), inc], check=True, env=dict(os.environ, LA_ASM_OPT=opt),
                   stdout=subprocess.DEVNULL)
    objs = os.path.join(OUT, "common.a")
    so = os.path.join(OUT, f"{name}.so")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
           f'-DLA_ASM_BODY_INC="{inc}"'] + [os.path.join(CSRC, s) for s in SRC] + ["-o", so]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return so


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    with ThreadPoolExecutor(4) as ex:
        for so in ex.map(build_one, sys.argv[1:]):
            print(so)
