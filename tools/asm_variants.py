"""Builds A/B variants of the hand-scheduled kernels: one library per generator-option setting, under build_variants/.

    python tools/asm_variants.py x_base=x64: x_nodma=x64:nodma f8_base=x64f8: ...
      name=x64:<opts>    gen_fwd_x64.py with LA_X64_OPT=<opts>
      name=x64f8:<opts>  gen_fwd_x64_fp8.py with LA_X64F8_OPT=<opts>
      name=x64d<D>:<opts> gen_fwd_x64.py with LA_X64_D=<D> (64, 96, 192, 256) LA_X64_OPT=<opts>
    (GPU box)  LITEATTENTION_AMD_LIB=$PWD/build_variants/<name>.so python tools/abl_bench.py
Ablation variants compute wrong results; they only price a component (HISTORY.md section 4).
"""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "liteattention_amd", "csrc")
OUT = os.path.join(ROOT, "build_variants")
sys.path.insert(0, ROOT)


def build_one(spec):
    name, _, opt = spec.partition("=")
    inc = os.path.join(OUT, f"{name}.inc")
    extra_env = {}
    if opt.startswith("x64f8:"):          # name=x64f8:<opts> -> gen_fwd_x64_fp8.py with LA_X64F8_OPT (bench with --dtype fp8)
        opt, gen, env_key, macro = opt[6:], "gen_fwd_x64_fp8.py", "LA_X64F8_OPT", "LA_X64F8_BODY_INC"
    elif opt.startswith("x64:"):
        opt, gen, env_key, macro = opt[4:], "gen_fwd_x64.py", "LA_X64_OPT", "LA_X64_BODY_INC"
    elif opt.startswith(("x64d64:", "x64d96:", "x64d192:", "x64d256:")):      # the other head dims of the bf16 generator (bench with tools/d64_bench.py / tools/d256_bench.py <D>)
        dim = opt[4:opt.index(":")]
        opt, gen, env_key, macro = opt[opt.index(":") + 1:], "gen_fwd_x64.py", "LA_X64_OPT", f"LA_X64_D{dim}_BODY_INC"
        extra_env["LA_X64_D"] = dim
    else:
        raise SystemExit(f"{spec}: options must start with x64: (bf16), x64d256: (bf16 head_dim 256) or x64f8: (fp8)")
    subprocess.run([sys.executable, os.path.join(CSRC, gen), inc], check=True, env=dict(os.environ, **{env_key: opt}, **extra_env),
                   stdout=subprocess.DEVNULL)
    # the other generated include must exist too (default options)
    so = os.path.join(OUT, f"{name}.so")
    from liteattention_amd.build import SOURCES, _build_record
    info = _build_record([inc], (), True)          # la_build_info() of the variant: its options, and wrong_results=1 for pricing bodies
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
           f'-D{macro}="{inc}"', f'-DLA_BUILD_INFO="{info}"'] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", so]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return so


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    with ThreadPoolExecutor(4) as ex:
        for so in ex.map(build_one, sys.argv[1:]):
            print(so)
