#!/usr/bin/env python
"""Host-side sanitizer runs of libmmd_amd.so (SURVEY §5 "race detection").  GPU AddressSanitizer / xnack+ code objects are
not available on the GPU pool, so this is the HOST half of the library, built and run in the build container (no GPU):

    python tools/host_sanitizers.py            # builds build_tmp/libmmd_amd_{asan,tsan}.so, runs the legs, writes profiles/r06_host_sanitizers.log

  leg 1  -Xarch_host -fsanitize=address,undefined (+ LeakSanitizer): tests/test_abi.py and tests/test_host_logic.py against that build,
         then every compute entry point called without a GPU -- each must come back with an error code and a message (hipErrorNoDevice),
         not a crash, and the creation paths must free what they allocated (the leak ADVICE r5 found in layered_create);
  leg 2  -Xarch_host -fsanitize=thread: the host-only entry points from 8 threads at once (mmd_pack_constraints, the thread-local
         mmd_last_error, failing mmd_unet_create calls) -- what planners.plan_concurrently does to the library from its worker threads.
Each leg is a subprocess with the sanitizer runtime preloaded; a sanitizer report fails the leg (exit code / report in the log)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = ["mmd_amd/csrc/unet.hip", "mmd_amd/csrc/unet_layers.hip", "mmd_amd/csrc/guide.hip", "mmd_amd/csrc/api.hip",
       "mmd_amd/csrc/multi_agent.hip", "mmd_amd/csrc/postprocess.hip"]
CLANG = "/opt/rocm/lib/llvm/bin/clang"

LEG = r'''
import ctypes as C, os, sys, threading
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np
from mmd_amd import _lib
_lib.LIB_PATH = sys.argv[2]
lib = _lib.load()
mode = sys.argv[3]
from mmd_amd import synth
sd = synth.synth_unet_state_dict(0)
sd1 = synth.synth_unet_state_dict(0, dim_mults=(1, 2, 4, 8))

def create(sdict, levels, flags=0):
    n = len(sdict)
    ptrs = (C.c_void_p * n)(*[v.ctypes.data for v in sdict.values()])
    numels = (C.c_int64 * n)(*[v.size for v in sdict.values()])
    h = C.c_void_p()
    opt = _lib.UnetOptions(flags, -1, 0, 0)
    rc = lib.mmd_unet_create(C.byref(h), 32, levels, 25, ptrs, numels, n, C.byref(opt), None)
    return rc, lib.mmd_last_error().decode()

def pack(seed):
    rng = np.random.default_rng(seed)
    G = 3
    n_pts = (C.c_int32 * G)(*[40, 7, 1953])
    arrs = []
    def pa(shape_fn):
        a = [np.ascontiguousarray(shape_fn(int(n)), dtype=np.float32) for n in n_pts]
        arrs.append(a)
        return (C.c_void_p * G)(*[x.ctypes.data for x in a])
    q = pa(lambda n: rng.uniform(-1, 1, (n, 2)))
    def ranges(n):
        t0 = rng.integers(0, 60, n)
        return np.stack([t0, t0 + rng.integers(1, 5, n)], 1)
    tr = pa(ranges)
    rad = pa(lambda n: np.full(n, 0.12))
    slots = (C.c_int32 * G)()
    assert lib.mmd_pack_constraints(G, n_pts, q, tr, rad, 64, None, 0, slots) == 0
    total = int(sum(slots))
    ell = np.zeros((total, 64, 4), np.float32)
    assert lib.mmd_pack_constraints(G, n_pts, q, tr, rad, 64, ell.ctypes.data, total, slots) == 0
    assert lib.mmd_pack_constraints(G, n_pts, q, tr, rad, 64, ell.ctypes.data, total - 1, slots) != 0   # too small: an error, no overrun
    assert b"slots" in lib.mmd_last_error()
    return total

if mode == "asan":
    # creation without a GPU: every path must fail cleanly and free what it allocated (fused, layered, option 1)
    for sdict, levels, flags in ((sd, 3, 0), (sd, 3, _lib.UNET_LAYERED), (sd1, 4, 0), (sd1, 4, _lib.UNET_LAYERED_VALU)):
        rc, msg = create(sdict, levels, flags)
        assert rc != 0 and msg, (rc, msg)
    assert lib.mmd_unet_num_tensors(12, 3) == -1 and b"unsupported" in lib.mmd_last_error()
    # every compute entry point with NULL / nonsense arguments: an error code and a message
    s, g = _lib.SamplerDesc(), _lib.GuideDesc()
    assert lib.mmd_unet_forward(None, None, 0, None, 4, None, 0, None) != 0
    assert lib.mmd_ddpm_step(None, C.byref(s), None, None, None, 1, 4, 0, None, 0, 0, None, 0, None) != 0
    assert lib.mmd_p_sample_loop(None, C.byref(s), C.byref(g), None, None, 1, 4, 25, 1, 1, None, 0, None, None, 0, None) != 0
    assert lib.mmd_p_sample_loop_ensemble(None, 0, None, 0, 1, 4, 25, 1, 1, None, 0, None) != 0
    assert lib.mmd_ddim_sample(None, C.byref(s), None, None, 0, None, None, None, 1, 4, 1, 0, None, None, 0, None) != 0
    assert lib.mmd_unet_workspace_bytes(None, 4) == 0 or True
    for seed in range(4):
        assert pack(seed) > 0
    print("ASAN_LEG_OK", flush=True)
else:
    errs = []
    def worker(k):
        try:
            for it in range(6):
                pack(100 * k + it)
                rc, msg = create(sd if k % 2 else sd1, 3 if k % 2 else 4, _lib.UNET_LAYERED if k % 3 == 0 else 0)
                assert rc != 0 and msg
                assert lib.mmd_unet_num_tensors(9 + 2 * k, 3) == -1 and b"unsupported" in lib.mmd_last_error()
        except Exception as e:      # noqa: BLE001
            errs.append((k, repr(e)))
    ts = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errs, errs
    print("TSAN_LEG_OK", flush=True)
'''


def build(tag, flags):
    out = os.path.join(ROOT, "build_tmp", f"libmmd_amd_{tag}.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-fno-slp-vectorize"] + [a for f in flags for a in ("-Xarch_host", f)] + SRC + ["-o", out]
    subprocess.run(cmd, check=True, cwd=ROOT)
    return out


def runtime(name):
    return subprocess.run([CLANG, f"--print-file-name=libclang_rt.{name}-x86_64.so"], capture_output=True, text=True, check=True).stdout.strip()


def main():
    log = []

    def run(title, cmd, env):
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=3000)
        text = r.stdout + r.stderr
        tail = [ln for ln in text.strip().splitlines() if ln.strip()]
        # memory errors / undefined behaviour / data races anywhere count; LEAK blocks count when a frame of the library is in the stack
        # (the uninstrumented interpreter and its extension modules leak at exit by design: those blocks are interpreter noise)
        blocks = text.split("\n\n")
        leaks_lib = [b for b in blocks if ("leak of" in b) and "libmmd_amd" in b]
        leaks_other = sum(1 for b in blocks if ("leak of" in b) and "libmmd_amd" not in b)
        bad = [ln for ln in tail if ("runtime error" in ln) or ("ERROR: AddressSanitizer" in ln) or ("WARNING: ThreadSanitizer" in ln)]
        bad += [b.strip().splitlines()[0] + "  [a frame of libmmd_amd in the stack]" for b in leaks_lib]
        ok_marks = [ln for ln in tail if ln.endswith("_LEG_OK") or " passed" in ln]
        log.append(f"== {title}\n   exit code {r.returncode}; memory-error / UB / race reports: {len(bad) - len(leaks_lib)}; leak blocks with a libmmd_amd frame: "
                   f"{len(leaks_lib)} (interpreter / extension-module leak blocks at exit, not counted: {leaks_other})\n   " + "\n   ".join(ok_marks[-3:]))
        if bad:
            log.append("   REPORTS:\n   " + "\n   ".join(bad[:40]))
        return r.returncode == 0 and not bad
    ok = True
    asan = build("asan", ["-fsanitize=address,undefined", "-fno-omit-frame-pointer"])
    # (python itself is not instrumented: leaks are reported for the library's frames only via the suppression of interpreter noise)
    supp = os.path.join(ROOT, "build_tmp", "lsan.supp")
    with open(supp, "w") as f:
        f.write("leak:libpython\nleak:_PyObject\nleak:PyMem\nleak:libtorch\nleak:libc10\nleak:numpy\nleak:libamdhip64\nleak:libhsa-runtime64\nleak:dl_open\nleak:_dl_\n")
    env = dict(os.environ, LD_PRELOAD=runtime("asan"), ASAN_OPTIONS="detect_leaks=1:halt_on_error=0:protect_shadow_gap=0",
               LSAN_OPTIONS=f"suppressions={supp}:print_suppressions=0", UBSAN_OPTIONS="print_stacktrace=1",
               MMD_AMD_LIB=asan, HIP_VISIBLE_DEVICES="")
    ok &= run("leg 1a: ASAN+UBSAN, error paths and host entry points without a GPU", [sys.executable, "-c", LEG, ROOT, asan, "asan"], env)
    ok &= run("leg 1b: ASAN+UBSAN, tests/test_abi.py + tests/test_host_logic.py on the instrumented build",
              [sys.executable, "-c", "import sys, os; sys.path.insert(0, sys.argv[1]); from mmd_amd import _lib; _lib.LIB_PATH = sys.argv[2]; "
               "import pytest; sys.exit(pytest.main(['-q', '-x', '-p', 'no:cacheprovider', os.path.join(sys.argv[1], 'tests', 'test_abi.py'), "
               "os.path.join(sys.argv[1], 'tests', 'test_host_logic.py')]))", ROOT, asan], env)
    tsan = build("tsan", ["-fsanitize=thread"])
    env = dict(os.environ, LD_PRELOAD=runtime("tsan"), TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0", HIP_VISIBLE_DEVICES="")
    ok &= run("leg 2: TSAN, host entry points from 8 threads", [sys.executable, "-c", LEG, ROOT, tsan, "tsan"], env)
    log.append("RESULT: " + ("all legs clean" if ok else "FAILED"))
    text = "\n".join(log) + "\n"
    print(text)
    with open(os.path.join(ROOT, "profiles", "r06_host_sanitizers.log"), "w") as f:
        f.write(__doc__.split("\n\n")[0] + "\n\n" + text)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
