"""Build libpsvae_hip.so for gfx950 with hipcc (explicit, in-tree; no JIT cache).

    python -m pseudo_speaker_vae_amd.build [--force]

The .so lands next to this file so it travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
REPO = os.path.dirname(PKG)
LIB = os.path.join(PKG, "libpsvae_hip.so")
SOURCES = ["gemm_f32.hip", "gemm_bf16.hip", "wgrad_bf16.hip", "elementwise.hip", "bottleneck.hip", "langevin.hip", "decode_fused.hip", "chain_bf16.hip", "classifier.hip", "heads_general.hip", "ipc_allreduce.hip", "psvae_api.hip",
           "psvae_api_bf16.hip"]
ARCH = "gfx950"


# kernels whose inline-asm register loads must never be spilled (hipcc would store the destination register before the
# data has landed and reload garbage): file -> substring of the mangled kernel names
# per-file compiler flags.  langevin.hip: no SLP vectorisation -- under plain -O3 hipcc packs adjacent scalar fp32 adds / muls into
# v_pk_* instructions, which cost the chain's MFMA loop time beside the MFMAs (cdna guide; measured 40.6 -> 40.1 ms per
# 1 M rows x 100 steps)
# gemm_f32.hip: the same for the epilogues beside the fp32 MFMA loops (cfg2 step 128.4 -> 127.5 us, three interleaved runs)
EXTRA_FLAGS = {"langevin.hip": ["-fno-slp-vectorize"], "gemm_f32.hip": ["-fno-slp-vectorize"]}

NO_SPILL = {}  # (none at present: the bf16 decode kernel that needed it was replaced by one without asm loads)


def _check_no_spill(src: str, remarks: str) -> None:
    import re
    name = None
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"VGPRs Spill: (\d+)", line)
        if m and name and NO_SPILL[src] in name and int(m.group(1)) > 0:
            raise RuntimeError(f"{src}: {name} spills {m.group(1)} vector registers; its asm loads' destinations may be "
                               f"among them (csrc/decode_fused.hip: the bf16 kernel's latent-row loads)")
    for line in remarks.splitlines():  # everything that is not a resource remark is a real diagnostic
        if "kernel-resource-usage" not in line and line.strip() and not line.startswith(" "):
            sys.stderr.write(line + "\n")


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: cannot build libpsvae_hip.so")


HASHFILE = LIB + ".srchash"


def _src_hash() -> str:
    """content hash of every source the library is built from (mtimes do not survive a repo snapshot)"""
    import hashlib
    h = hashlib.sha256()
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(REPO, "include", "psvae.h")]
    for d in deps:
        h.update(os.path.basename(d).encode())
        h.update(open(d, "rb").read())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def _stale():
    if not (os.path.exists(LIB) and os.path.exists(HASHFILE)):
        return True
    return open(HASHFILE).read().strip() != _src_hash()


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    objs = []
    bdir = os.path.join(PKG, "build")
    os.makedirs(bdir, exist_ok=True)
    common = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(REPO, "include")]
    procs = []
    for src in SOURCES:
        obj = os.path.join(bdir, src.replace(".hip", ".o"))
        cmd = common + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if src in NO_SPILL:  # resource remarks on stderr, checked below
            cmd.insert(-4, "-Rpass-analysis=kernel-resource-usage")
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd, stderr=subprocess.PIPE if src in NO_SPILL else None, text=True), src))
        objs.append(obj)
    for p, src in procs:
        err = p.communicate()[1] if src in NO_SPILL else None
        if p.wait() != 0:
            if err:
                sys.stderr.write(err)
            raise RuntimeError(f"hipcc failed on {src}")
        if err is not None:
            _check_no_spill(src, err)
    tmp = LIB + f".tmp{os.getpid()}"
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(tmp, LIB)  # atomic: a concurrent loader never sees a half-written library
    with open(HASHFILE, "w") as f:
        f.write(_src_hash())
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
