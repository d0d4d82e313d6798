"""C-ABI library: loads without a GPU, exports every symbol include/topolow_relax.h declares,
host-side helpers agree with the oracle / with first principles, and compute entry points
refuse to run without a device (no CPU fallback)."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from oracle import topolow_oracle as orc
from topolow_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "relax_golden.json")))


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_library_exports_every_declared_symbol():
    lib = _native.load()
    header = open(os.path.join(ROOT, "include", "topolow_relax.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    names = set(re.findall(r"\b(topolow_[a-z0-9_]+)\s*\(", header))
    assert len(names) >= 25
    for nm in sorted(names):
        assert hasattr(lib, nm), f"{nm} declared in include/topolow_relax.h but not exported"


def test_target_encoding_roundtrip():
    rng = np.random.default_rng(0)
    for t in list(rng.uniform(0, 50, 200)) + [0.0, 1e-30, 3.4e38, 1e300, -2.5]:
        for code in (0, 1, -1):
            w = _native.encode_target(t, code)
            v, c = _native.decode_target(w)
            assert c == code
            if abs(t) < 3e38:
                assert v == pytest.approx(t, rel=3e-7, abs=1e-37)
            else:
                assert np.isfinite(v)   # huge finite targets stay measured
    for bad in (np.inf, -np.inf, np.nan):
        v, c = _native.decode_target(_native.encode_target(bad, 1))
        assert v == np.inf and c == -1  # not finite -> unmeasured = (+Inf, "<"): never a spring


@pytest.mark.parametrize("n,stages", [(10000, 4), (10001, 16), (257, 8), (50000, 64), (33, 4), (7, 16)])
def test_slab_plan_covers_every_column_once(n, stages):
    n4 = (n + 3) // 4 * 4
    for it in range(5):
        plan = _native.slab_plan(n, stages, 1234, it)
        assert 1 <= len(plan) <= stages
        cover = np.zeros(n4, dtype=int)
        for b0, e0, b1, e1 in plan:
            assert b0 % 4 == 0 and e0 % 4 == 0 and b1 % 4 == 0 and e1 % 4 == 0
            cover[b0:e0] += 1
            cover[b1:e1] += 1
        assert np.all(cover == 1)
    assert not np.array_equal(_native.slab_plan(n, stages, 1234, 0), _native.slab_plan(n, stages, 1234, 1)) \
        or n < 64


def test_slab_stage_policy():
    # k / stages <= min(3, ndim): one sweep per iteration once k <= 3 (ndim >= 3)
    assert _native.slab_stages_for_k(0.1, 5) == 1 and _native.slab_stages_for_k(3.0, 5) == 1
    assert _native.slab_stages_for_k(3.01, 5) == 2 and _native.slab_stages_for_k(5.0, 5) == 2
    assert _native.slab_stages_for_k(10.0, 3) == 4
    assert _native.slab_stages_for_k(14.76, 5) == 8 and _native.slab_stages_for_k(30.0, 5) == 16
    assert _native.slab_stages_for_k(5.0, 1) == 8 and _native.slab_stages_for_k(5.0, 2) == 4     # stability: k / S < 2 ndim
    # while the layout unfolds (first 8 iterations) never fewer than 16 stages
    assert _native.slab_stages_at(0, 5.0, 5) == 16 and _native.slab_stages_at(7, 5.0, 5) == 16
    assert _native.slab_stages_at(8, 5.0, 5) == 2 and _native.slab_stages_at(3, 100.0, 5) == 64


@pytest.mark.parametrize("n", [2, 3, 5, 8, 33, 64, 101])
def test_gs_pair_order_is_a_permutation_of_all_pairs_in_disjoint_rounds(n):
    for it in range(3):
        pairs = _native.gs_pair_order(n, 99, it)
        key = {(min(a, b), max(a, b)) for a, b in pairs}
        assert len(key) == n * (n - 1) // 2 and all(a != b for a, b in pairs)
        per_round = n // 2
        for r in range(0, len(pairs), per_round):
            pts = pairs[r:r + per_round].ravel()
            assert len(set(pts.tolist())) == len(pts)   # a round's pairs are disjoint
    if n > 3:
        assert not np.array_equal(_native.gs_pair_order(n, 99, 0), _native.gs_pair_order(n, 99, 1))


@pytest.mark.parametrize("case", GOLD["G4_controller"], ids=lambda c: f"{c['name']}-w{c['window']}")
def test_device_controller_code_matches_golden_and_oracle(case):
    maes = [float(m) if m not in ("nan",) else float("nan") for m in case["maes"]]
    mine = _native.controller_script(maes, case["iters"], case["ks"], case["k0"], case["window"], case["eps"])
    ref = orc.controller_script(maes, case["iters"], case["ks"], case["k0"], case["window"], case["eps"])
    for key in ("stopped_at", "best_iter", "best_mae", "best_k"):
        assert mine[key] == ref[key] == case["expect"][key]
    assert list(mine["snapshots"]) == list(ref["snapshots"])


@pytest.mark.skipif(_has_gpu(), reason="checks the no-device error path")
def test_no_cpu_fallback_without_device():
    D = np.array([[np.inf, 1.0], [1.0, np.inf]])
    T = np.zeros((2, 2), np.int32)
    with pytest.raises(_native.NativeError) as ei:
        _native.optimize_layout_exact_arrays(np.zeros((2, 2)), D, T, [1, 1], [0], [1], [1.0], [0], 5, 1.0,
                                             0.1, 0.1, 1e-4, 5, 3, seed=1)
    assert ei.value.code == _native.ERR_NO_DEVICE
    with pytest.raises(_native.NativeError) as ei:
        _native.est_distances(np.zeros((3, 2)))
    assert ei.value.code == _native.ERR_NO_DEVICE


def test_too_few_points_message():
    with pytest.raises(_native.NativeError, match="Need at least 2 points for embedding"):
        _native.optimize_layout_exact_arrays(np.zeros((1, 2)), np.zeros((1, 1)), np.zeros((1, 1), np.int32),
                                             [0], [], [], [], [], 5, 1.0, 0.1, 0.1, 1e-4, 5, 3, seed=1)


def test_struct_layout_matches_the_header(tmp_path):
    """The ctypes twins of topolow_problem / topolow_result / topolow_options must have the C sizes."""
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "topolow_relax.h"\nint main(void){printf("%zu %zu %zu\\n",'
                   'sizeof(topolow_problem), sizeof(topolow_result), sizeof(topolow_options));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    sizes = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert sizes == [C.sizeof(_native.TopolowProblem), C.sizeof(_native.TopolowResult),
                     C.sizeof(_native.TopolowOptions)]


def test_stage_kernel_awaits_its_lds_transfers_past_exactly_the_younger_loads():
    """The stage kernel issues its direct-to-LDS transfers from inline asm and, before the chunk
    barrier, waits for them with `s_waitcnt vmcnt(N)`, N = the target loads issued since (vmcnt counts
    in issue order).  N is a template constant; the loads are the compiler's.  If the compiler ever
    dropped or merged one of them the wait would be too weak (a transfer could still be in flight at
    the barrier), so the ISA of every instantiation is checked: between the last transfer of the loop
    body and the wait there must be exactly N `buffer_load_dwordx4`."""
    import subprocess
    csrc = os.path.join(ROOT, "topolow_amd", "csrc")
    subprocess.run(["make", "-C", csrc, "asm"], check=True, capture_output=True)
    text = open(os.path.join(csrc, "topolow_relax.gfx950.s")).read()
    names = re.findall(r"^(_ZN7topolow22slab_stage_pipe_kernel\w+):", text, re.M)
    # coordinates x {f32, f64} x {threshold, threshold-free} + the fp32 instances that also reduce the MAE
    assert len(names) == 72      # 12 coordinate counts (1..10, 12, 16)
    for name in names:
        start = text.index("\n" + name + ":")
        body = text[start:text.index(".Lfunc_end", start)].split("\n")
        waits = [q for q, l in enumerate(body) if "s_waitcnt vmcnt(" in l and "ASMSTART" in body[q - 1]]
        counts = [int(re.search(r"vmcnt\((\d+)\)", body[q]).group(1)) for q in waits]
        assert len(waits) == 2 and counts[0] == 0 and counts[1] > 0, (name, counts)   # prologue, loop
        q = waits[1]
        dma = max(i for i in range(q) if "global_load_lds_dwordx4" in body[i])
        assert dma > waits[0]                     # the loop's own transfer, not the prologue's
        younger = sum("buffer_load_dwordx4" in l for l in body[dma:q])
        assert younger == counts[1], (name, younger, counts[1])
        # and no instantiation may spill registers (the threshold-carrying fp32 instance at ndim 10
        # once did, inside the pair loop: 9x slower)
        tail = text[text.index(".Lfunc_end", start):][:3000]
        assert re.search(r"; ScratchSize: (\d+)", tail).group(1) == "0", name


def test_symmetric_sweep_instances_fit_their_register_budget():
    """Every instance of the symmetric sweep the library launches (ndim 2..6 x {threshold-free, threshold} x {plain,
    ERR}) must run without scratch and with at least two waves per SIMD; its column reduction must be the
    3 x (2 x ndim) single-instruction DPP adds per half tile the kernel's header describes (the optimiser once turned
    them into v_mov_dpp + v_pk_add pairs)."""
    import subprocess
    csrc = os.path.join(ROOT, "topolow_amd", "csrc")
    subprocess.run(["make", "-C", csrc, "asm"], check=True, capture_output=True)
    text = open(os.path.join(csrc, "topolow_relax.gfx950.s")).read()
    names = re.findall(r"^(_ZN7topolow17symm_sweep_kernelILi(\d+)ELb([01])ELb([01])E\w+):", text, re.M)
    assert len(names) == 20
    for name, dim, thr, err in names:
        start = text.index("\n" + name + ":")
        end = text.index(".Lfunc_end", start)
        tail = text[end:][:3000]
        assert re.search(r"; ScratchSize: (\d+)", tail).group(1) == "0", name
        assert int(re.search(r"; Occupancy: (\d+)", tail).group(1)) >= 2, name
        body = text[start:end]
        # three DPP steps on ndim values per half tile (csrc/relax_symm.h: sym_col_reduce)
        assert body.count("v_add_f32_dpp") % (3 * int(dim)) == 0 and "v_add_f32_dpp" in body, name
        # DPP hazard: a VGPR written by a vector instruction may be read through DPP only two wait states later; the
        # adds are inline asm, so the compiler neither knows nor pads -- no instruction among the two before a DPP add
        # may write the register it reads through DPP (an s_nop counts as wait states)
        lines = [ln.strip() for ln in body.split("\n") if ln.strip() and not ln.strip().startswith((";", "."))]
        for k, ln in enumerate(lines):
            m = re.match(r"v_add_f32_dpp v\d+, v(\d+), v\d+", ln)
            if not m:
                continue
            src = int(m.group(1))
            waits = 0
            for prev in reversed(lines[max(0, k - 4):k]):
                if waits >= 2:
                    break
                nop = re.match(r"s_nop (\d+)", prev)
                if nop:
                    waits += int(nop.group(1)) + 1
                    continue
                w = re.match(r"v_\w+ v(\d+)|v_\w+ v\[(\d+):(\d+)\]", prev)
                if w:
                    lo = int(w.group(1) or w.group(2))
                    hi = int(w.group(1) or w.group(3))
                    assert not (lo <= src <= hi), (name, prev, ln)
                waits += 1


def test_f64_symmetric_sweep_instances_run_without_scratch():
    """csrc/relax_symm64.h: all twenty instances (ndim 2..6 x {threshold-free, threshold} x {plain, ERR}) without
    scratch; two waves per SIMD up to ndim 5 for the plain ones (the ERR instances carry 32 more registers of delta
    words: one wave from ndim 5, ndim 4 with thresholds).  (The kernel once kept every pair's
    dx and factor alive to the end of the tile -- 370 registers at ndim 2, scratch from ndim 4 -- because a divergent
    store split the tile into basic blocks and nothing ordered the pure arithmetic: the tile is one block now and the
    sums are pinned after every four pairs.)"""
    import subprocess
    csrc = os.path.join(ROOT, "topolow_amd", "csrc")
    subprocess.run(["make", "-C", csrc, "asm"], check=True, capture_output=True)
    text = open(os.path.join(csrc, "topolow_relax.gfx950.s")).read()
    names = re.findall(r"^(_ZN7topolow19symm64_sweep_kernelILi(\d+)ELb([01])ELb([01])E\w+):", text, re.M)
    assert len(names) == 20
    for name, dim, thr, err in names:
        start = text.index("\n" + name + ":")
        end = text.index(".Lfunc_end", start)
        tail = text[end:][:3000]
        assert re.search(r"; ScratchSize: (\d+)", tail).group(1) == "0", name
        two = int(dim) <= 3 or (int(dim) == 4 and not (thr == "1" and err == "1")) or (int(dim) == 5 and err == "0")
        assert int(re.search(r"; Occupancy: (\d+)", tail).group(1)) >= (2 if two else 1), name
        body = text[start:end]
        assert body.count("s_cbranch") <= 19, (name, body.count("s_cbranch"))     # loops and guards only (13-18): the tile has no branch
