"""Build libyond_hip.so for gfx950 with hipcc (cross-compiles without a GPU): every csrc/*.hip to an object file in
parallel (csrc/.obj/, one hipcc per source, re-used while newer than the source and every header), then one link."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, ".obj")
LIB = os.path.join(HERE, "libyond_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: every float32/float64 rounding point of the NumPy-staged reference is reproduced;
# fused multiply-adds appear only where the source says fma()/fmaf() or in the MFMA instructions.
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "yond_hip.h"),
                                                   os.path.join(HERE, "..", "include", "yond_hip_experiments.h")]


def _obj(src, extra_flags=()):
    # (experiment builds with extra flags keep their objects apart from the product's)
    import hashlib
    sub = hashlib.sha256(" ".join(extra_flags).encode()).hexdigest()[:8] if extra_flags else "product"
    return os.path.join(OBJ, sub, os.path.basename(src)[:-4] + ".o")


def _deps(src):
    """src + the headers it includes (transitively; quoted includes resolved against csrc/ and include/)."""
    import re
    seen, todo = [], [src]
    while todo:
        f = todo.pop()
        if f in seen or not os.path.exists(f):
            continue
        seen.append(f)
        for inc in re.findall(r'#include\s+"([^"]+)"', open(f).read()):
            todo.append(os.path.normpath(os.path.join(os.path.dirname(f), inc)))
    return seen


def _dep_hash(src, extra_flags=()):
    """Content hash of what an object file is compiled from: the flags, the source and every header it includes."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS + list(extra_flags)).encode())
    for f in sorted(_deps(src)):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _stale(obj, src, extra_flags=()):
    """An object is reused only if the stamp beside it holds the hash of the flags and of its source + headers (content, not
    modification times: a changed FLAGS list or an edited header under an old mtime recompiles)."""
    stamp = obj + ".hash"
    return not (os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read().strip() == _dep_hash(src, extra_flags))


def _write_resource_report(remarks, path):
    import json
    import re
    out, cur = [], None
    show, held = False, []                       # inside a warning / error / note (its excerpt lines follow); "In file included from" lines held for it
    for line in remarks.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            out.append(cur)
        if "remark:" in line:
            show, held = False, []
        elif re.search(r"(warning|error|note):", line) or re.match(r"^\d+ (warning|error)s? generated", line.strip()):
            # compiler diagnostics go through wherever they stand (they used to be dropped behind the first kernel's remarks)
            show = True
            sys.stderr.write("".join(h + "\n" for h in held) + line + "\n")
            held = []
        elif line.startswith("In file included from"):
            held.append(line)
        elif show and line.strip():
            sys.stderr.write(line + "\n")        # the diagnostic's source excerpt / caret
        if "remark:" not in line or cur is None:
            continue
        for key, tag in (("vgprs", "VGPRs:"), ("vgpr_spill", "VGPRs Spill:"), ("sgpr_spill", "SGPRs Spill:"), ("scratch", "ScratchSize [bytes/lane]:"),
                         ("lds", "LDS Size [bytes/block]:")):
            m = re.search(re.escape(tag) + r"\s*(\d+)", line)
            if m and "remark:     " + tag in line:
                cur[key] = int(m.group(1))
    with open(path, "w") as f:
        json.dump(out, f)


def resource_report():
    """[{name (mangled), vgprs, vgpr_spill, sgpr_spill, scratch, lds}] of every kernel of the product build (from the objects' reports)."""
    import json
    rep, missing = [], 0
    for s in sources():
        p = _obj(s) + ".res.json"
        if os.path.exists(p):
            rep += json.load(open(p))
        else:
            missing += 1
    if missing and os.path.exists(RES_STAMP) and not needs_build():
        # a shipped library (reused by its stamp) without its object directory: the report written when it was linked
        return json.load(open(RES_STAMP))
    return rep


STAMP = os.path.join(HERE, "libyond_hip.stamp")      # sha256 of (flags, sources, headers) the library was built from
RES_STAMP = os.path.join(HERE, "libyond_hip.res.stamp")   # the kernels' resource report (JSON) as of that link: travels with the library


def source_hash(extra_flags=()):
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS + list(extra_flags)).encode())
    for f in sorted(sources() + [os.path.normpath(x) for x in headers()]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def needs_build():
    """True unless the library exists and was built from exactly these sources (content hash, not mtimes: a snapshot or
    a checkout does not keep them)."""
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return True
    return open(STAMP).read().split()[0] != source_hash()


def build_lib(force=False, verbose=True, extra_flags=(), lib=None):
    """Compile what is out of date (force=True: everything) and link.  Returns (path, mode) with mode one of
    'reused' (the library's stamp equals the hash of the sources), 'compiled N of M sources'."""
    lib = lib or LIB
    if not force and lib == LIB and not needs_build():
        if verbose:
            print(f"yond_public_amd.build: {LIB} matches the hash of every source: reused", flush=True)
        return LIB, "reused"
    os.makedirs(os.path.dirname(_obj("x.hip", extra_flags)), exist_ok=True)
    todo = [s for s in sources() if force or _stale(_obj(s, extra_flags), s, extra_flags)]

    def cc(src):
        obj = _obj(src, extra_flags)
        cmd = [HIPCC] + FLAGS + list(extra_flags) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        if os.path.exists(obj + ".hash"):
            os.remove(obj + ".hash")
        # the compiler's per-kernel resource remarks ride along: registers, spills, scratch of every kernel -> <obj>.res.json
        # (tests/test_build_report.py: no kernel the networks launch may spill)
        r = subprocess.run(cmd + ["-Rpass-analysis=kernel-resource-usage"], stderr=subprocess.PIPE, text=True)
        if r.returncode:
            sys.stderr.write(r.stderr)
            raise subprocess.CalledProcessError(r.returncode, cmd)
        _write_resource_report(r.stderr, obj + ".res.json")
        with open(obj + ".hash", "w") as f:
            f.write(_dep_hash(src, extra_flags) + "\n")
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(todo)))) as ex:
        list(ex.map(cc, todo))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + [_obj(s, extra_flags) for s in sources()]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    mode = f"compiled {len(todo)} of {len(sources())} sources"
    if lib == LIB:
        with open(STAMP, "w") as f:
            f.write(f"{source_hash(extra_flags)} {mode}\n")
        import json
        with open(RES_STAMP, "w") as f:
            json.dump(resource_report(), f)
    if verbose:
        print(f"yond_public_amd.build: {mode}, linked {lib}", flush=True)
    return lib, mode


EXP_LIB = os.path.join(HERE, "..", "tools", "probe", "libyond_exp.so")


def build_experiments(extra=()):
    """The experiment library (tools/probe/libyond_exp.so): -DYOND_EXPERIMENTS adds the environment switches of the launch
    heuristics and the measured-and-dropped alternatives (one-pass NLE kernels); use it with YOND_HIP_LIB=<path>."""
    os.makedirs(os.path.dirname(EXP_LIB), exist_ok=True)
    return build_lib(extra_flags=["-DYOND_EXPERIMENTS"] + list(extra), lib=os.path.normpath(EXP_LIB))


if __name__ == "__main__":
    if "--experiments" in sys.argv:
        print(build_experiments())
    else:
        print(build_lib(force="--force" in sys.argv)[1])
