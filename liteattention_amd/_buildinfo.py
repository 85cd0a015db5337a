"""What the native library was built from, computed the same way at build time (``build.py`` bakes it into ``la_build_info()``) and at
load time (``_cabi.load()`` compares): sha256 over the kernel / API sources, the generators and the public header. The generated
``*_body.inc`` / ``*_consts.h`` are outputs of the generators and not part of it."""
from __future__ import annotations

import hashlib
import os
import re
from typing import Dict, Optional

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
PUBLIC_HEADER = os.path.join(os.path.dirname(PKG_DIR), "include", "lite_attention_amd.h")
SOURCES = ["la_fwd_kernel_v2.hip", "la_fwd_kernel_x64.hip", "la_prep_fp8.hip", "la_fwd_kernel_x64_fp8.hip", "la_aux_kernels.hip",
           "la_api.hip"]
HEADERS = ["la_kernel_params.h", "la_tiles.h", "la_fwd_common.h", "gen_fwd_x64.py", "gen_fwd_x64_fp8.py", "gen_epilogue.py"]


def source_hash() -> Optional[str]:
    """sha256[:16] of the build inputs, or None when the sources are not beside the package (a deployed library without its tree)."""
    h = hashlib.sha256()
    for path in [os.path.join(CSRC, n) for n in SOURCES + HEADERS] + [PUBLIC_HEADER]:
        if not os.path.exists(path):
            return None
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]


def parse(info: str) -> Dict[str, str]:
    """"abi=7;src=...;variant=0;wrong_results=0;opts=..." -> dict (opts may itself hold ';'-free text only)."""
    out = {}
    for part in info.split(";"):
        k, _, v = part.partition("=")
        out[k] = v
    return out


def record_in_file(lib_path: str) -> Optional[Dict[str, str]]:
    """The build record of a library file WITHOUT loading it (the string la_build_info() returns sits in .rodata)."""
    try:
        with open(lib_path, "rb") as f:
            m = re.search(rb"src=([0-9a-z]{1,16});variant=([01]);wrong_results=([01]);opts=([^\0]*)\0", f.read())
    except OSError:
        return None
    if not m:
        return None
    return {"src": m.group(1).decode(), "variant": m.group(2).decode(), "wrong_results": m.group(3).decode(), "opts": m.group(4).decode()}
