"""Static tripwires for the hand-counted waits of the direct-A kernels (conv_f16x3_da.hip, conv_f16x3_dag.hip).

Their weight rings are loaded by inline asm and waited for with `s_waitcnt vmcnt(N)`, N counted by hand from what is
in the wave's in-order memory queue: 2 per ring refill and `raw_ops` per input prefetch batch.  Two things the
compiler could do silently would break that (stale fragments, only when memory is slow):
  * touch the destination registers of an asm load before the next hand-written wait (it once copied them there), and
  * emit a different number of vector loads for the input prefetch than `raw_ops` says (merging or splitting loads).
Both are checked here on the generated assembly (hipcc cross-compiles without a GPU)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


# every translation unit this file audits, with its extra flags: compiled to assembly side by side on first use (five hipcc runs
# of 10 - 70 s each; in a row they were most of the CPU suite's time)
_UNITS = {"conv_f16x3_da.hip": ("-DKX_DA_AUDIT",), "conv_f16x3_da_p1.hip": ("-DKX_DA_AUDIT",), "conv_f16x3_da_w2.hip": ("-DKX_DA_AUDIT",),
          "conv_f16x3_da_s16.hip": ("-DKX_DA_AUDIT",), "conv_f16x3_dag.hip": (), "conv_f16x3_da_pre.hip": (), "conv_f16x3_dapn.hip": (),
          "conv_f16x3_da_f8.hip": ()}
_ASM_JOBS = {}


def _asm_one(src, flags, out_dir):
    out = os.path.join(out_dir, os.path.basename(src) + ".s")
    subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "--cuda-device-only", "-S", *flags,
                    os.path.join(ROOT, "kokorox_amd", "csrc", src), "-o", out], check=True, capture_output=True)
    return open(out).read()


def _asm(src, tmp_path, *flags):
    """assembly text of `src` (the first call starts all five compilations in parallel; each test waits for its own)"""
    assert tuple(flags) == _UNITS[src], (src, flags)
    if not _ASM_JOBS:
        import tempfile
        from concurrent.futures import ThreadPoolExecutor
        out_dir = tempfile.mkdtemp(prefix="kx_asm_audit_")
        ex = ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1))
        for u, fl in _UNITS.items():
            _ASM_JOBS[u] = ex.submit(_asm_one, u, fl, out_dir)
        ex.shutdown(wait=False)
    return _ASM_JOBS[src].result()


def _kernels(text):
    """name -> list of instruction lines of every kernel function in the assembly"""
    ks, cur, name = {}, None, None
    for ln in text.splitlines():
        m = re.match(r"^(_ZN2kx\w+):", ln)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if ln.startswith(".Lfunc_end"):
                ks[name], cur = cur, None
            else:
                cur.append(ln)
    return ks


def _regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def _audit_no_touch_before_wait(lines):
    """No instruction outside the hand-written asm may name the destination registers of an asm `global_load_dword[x4]`
    while that load can still be in flight.  Vector-memory operations of a wave complete in order, so a load with k younger
    vector-memory operations behind it has landed at the first `s_waitcnt vmcnt(N)` with N <= k (text order; the branches
    of the wait ladder all precede the use); waits that name only lgkmcnt retire nothing."""
    pending, in_asm, bad, vm = [], False, [], 0
    for ln in lines:
        s = ln.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        toks = re.findall(r"v\[\d+:\d+\]|v\d+\b", s)
        if s.startswith("s_waitcnt"):
            m = re.search(r"vmcnt\((\d+)\)", s)
            if m:
                n = int(m.group(1))
                pending = [(regs, at) for regs, at in pending if n > vm - at]
            continue
        if re.match(r"(global|buffer|flat|scratch)_(load|store|atomic)", s):
            vm += 1
            if in_asm and (s.startswith("global_load_dword") or s.startswith("buffer_load_dwordx4")) and toks:  # ring (dwordx4) and input prefetch (dword) loads
                pending.append((_regs(toks[0]), vm))
                continue
        if in_asm:
            continue
        used = set().union(*[_regs(t) for t in toks]) if toks else set()
        if any(used & regs for regs, _ in pending):
            bad.append(s)
    return bad


def _audit_spills(name, lines):
    """No spill between the first and the last MFMA (a spill reload is a vector-memory operation the hand-counted waits do not know of);
    the epilogue may park ONE 16-byte value (round 5: the 2 x 2 forms carry the streaming and the cacheable wide store, twice each)."""
    mfi = [i for i, ln in enumerate(lines) if "v_mfma" in ln]
    assert not any("scratch_" in ln for ln in lines[mfi[0]:mfi[-1] + 1]), f"{name} spills inside the main loop"
    assert sum(1 for ln in lines if "scratch_" in ln) <= 2, f"{name}: more than one spilled value"


def _prefetch_batches(lines):
    """Sizes of the runs of scalar-dword vector loads that follow an s_barrier of the MAIN LOOP (the input prefetch of a
    chunk boundary).  A run ends at the next MFMA or the next barrier, whichever comes first in the text: the compiler may
    rotate the loop and place the latch (barrier + prefetch) ahead of the loop body, right behind the prologue's own
    barrier + prefetch.  The epilogue's barrier is told apart by the buffer_* accesses that follow it (the main loop has
    none)."""
    sizes, i = [], 0
    while i < len(lines):
        if lines[i].strip() == "s_barrier":
            n, j, epilogue = 0, i + 1, False
            while j < len(lines) and j < i + 4000 and "v_mfma" not in lines[j] and lines[j].strip() != "s_barrier":
                if re.match(r"\s*global_load_dword\s", lines[j]):
                    n += 1
                if re.match(r"\s*buffer_(load|store)", lines[j]):
                    epilogue = True
                    break
                j += 1
            if n >= 8 and not epilogue:
                sizes.append(n)
            i = j
        else:
            i += 1
    return sizes


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("src", ["conv_f16x3_da.hip", "conv_f16x3_da_p1.hip", "conv_f16x3_da_w2.hip", "conv_f16x3_da_s16.hip"])  # f16x3, reduced precision, 2 x 2 waves, 16x16x32
def test_direct_a_conv_assembly(tmp_path, src):
    ks = _kernels(_asm(src, tmp_path, "-DKX_DA_AUDIT"))
    assert ks, "no kernel found"
    # every instantiation launch_da_ntt can select, both tile widths; the four unrolled 256-column forms of the 2 x 2 layout
    # (the reduced-precision unit holds every form twice since round 5: f16 and bf16 operands)
    assert len(ks) == {"conv_f16x3_da_w2.hip": 4, "conv_f16x3_da_s16.hip": 4, "conv_f16x3_da_p1.hip": 28}.get(src, 14), sorted(ks)
    for name, lines in ks.items():
        _audit_spills(name, lines)
        bad = _audit_no_touch_before_wait(lines)
        assert not bad, f"{name}: ring registers touched before their wait: {bad[:3]}"
        m = re.search(r"da_kernelILi(\d+)ELi(\d+)ELi(\d+)ELb[01]E", name)
        act, kt, ntt = (int(x) for x in m.groups())
        # elements per lane and chunk: BN + 128 staged columns in the run-time form, BN + 64 (last block split over the four
        # waves by channel) in the unrolled forms
        hu = 8 * (ntt // 4) + 4 if kt > 0 else (32 * ntt + 128) // 128 * 8
        if ntt == 6:
            hu = 16  # (the 192-column tile of the S16 form: a window of four whole 64-column blocks, no split block)
        want = hu + 3 + (1 if act == 2 else 0)  # raw_ops with the three InstanceNorm parameter loads present
        sizes = _prefetch_batches(lines)
        assert sizes and all(sz == want for sz in sizes), f"{name}: input prefetch is {sizes} loads, raw_ops assumes {want}"
        # instruction budget of the unrolled main loops (text between the first and the last MFMA of the chunk body): vector
        # instructions per MFMA.  The matrix pipe hides ~5 issue slots per 32-cycle MFMA (MI355X_MICROARCH.md); the k = 11
        # form must stay well inside that, and none of the forms may quietly grow (round 3: 2.1 / 3.3 / 7.3 at 256 columns)
        if kt > 0 and src != "conv_f16x3_da_p1.hip":
            mf = [i for i, ln in enumerate(lines) if "v_mfma" in ln]
            body = lines[mf[0]:mf[-1] + 1]
            valu = sum(1 for ln in body if re.match(r"\s*v_(?!mfma)", ln))
            budget = {11: 3.0, 7: 4.5, 3: 9.5}[kt]
            assert valu / len(mf) <= budget, f"{name}: {valu / len(mf):.2f} vector instructions per MFMA in the main loop (budget {budget})"


def _f8_ages(kt, rh, rc, raw):
    """F8Sched::age of conv_f16x3_da.hip restated: the wait of every pair-step of a super-chunk in the steady state = vector-memory
    operations issued since the refill of the slot(s) it hands on (program order: [p == HS: input prefetch] [wait] .. [hi refill: 2]
    [cross refill: 4]; the input prefetch behind barrier 3)."""
    hs, ng = (kt - 1) // 2, (kt + 1) // 4
    nc = 2 * ng

    def cross_of(p):
        if p & 1:
            return -1
        if p <= 2 * (ng - 1):
            return p // 2
        if p >= kt - 1 - 2 * (ng - 1):
            return nc - 1 - (kt - 1 - p) // 2
        return -1

    n, st_h, st_x, waits = 0, {}, {}, []
    for sc in range(2):
        for p in range(kt):
            if p == hs:
                n += raw
            if sc == 1:
                a = n - st_h[kt + p]
                c = cross_of(p)
                if c >= 0:
                    a = min(a, n - st_x[nc + c])
                waits.append(min(a, 63))
            n += 2
            st_h[sc * kt + p + rh if p + rh <= kt - 1 else (sc + 1) * kt + p % rh] = n
            c = cross_of(p)
            if c >= 0:
                n += 4
                st_x[sc * nc + c + rc] = n
        n += raw
    return waits


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_f16f8_conv_assembly(tmp_path):
    """The f16f8 forms of the 16x16x32 loop (conv_f16x3_da_f8.hip): their rings are buffer loads in inline asm with waits that are
    compile-time constants (F8Sched::age).  Checked on the generated code: no spill inside the main loop (a spill reload is a vector-
    memory operation that drains the ring), the input prefetch is raw_ops loads, the waits of a super-chunk are those of an
    independent restatement of the schedule, no ring register is touched before its wait, and the MFMA mix is 2 f16 per block plus
    2 scaled per block of a cross-term pair-step."""
    ks = _kernels(_asm("conv_f16x3_da_f8.hip", tmp_path))
    assert len(ks) == 6, sorted(ks)
    for name, lines in ks.items():
        m = re.search(r"da_kernelILi(\d+)ELi(\d+)ELi(\d+)ELb0ELb0ELb1ELb0ELb0ELb1E", name)
        assert m, name
        act, kt, ntt = (int(x) for x in m.groups())
        assert act == 2 and kt in (3, 7, 11) and ntt in (4, 6)
        mf = [i for i, ln in enumerate(lines) if "v_mfma" in ln]
        body = lines[mf[0]:mf[-1] + 1]
        assert not any("scratch_" in ln for ln in body), f"{name} spills inside the main loop"
        bad = _audit_no_touch_before_wait(lines)
        assert not bad, f"{name}: ring registers touched before their wait: {bad[:3]}"
        hu = 16 if ntt == 6 else 12
        raw = hu + 4
        # the input prefetch behind a barrier of the loop body: the forms' raw loads are scalar-dword BUFFER loads (the three
        # InstanceNorm parameters and alpha: global loads)
        sizes, i = [], 0
        while i < len(body):
            if body[i].strip() == "s_barrier":
                n, j = 0, i + 1
                while j < len(body) and "v_mfma" not in body[j] and body[j].strip() != "s_barrier":
                    n += 1 if re.match(r"\s*(buffer|global)_load_dword\s", body[j]) else 0
                    j += 1
                sizes.append(n)
                i = j
            else:
                i += 1
        sizes = [n for n in sizes if n >= 8]  # (barrier 2 is followed by no prefetch)
        assert sizes and all(sz == raw for sz in sizes), f"{name}: input prefetch is {sizes} loads, raw_ops assumes {raw}"
        # the hand-written waits of the loop body, in text order (they sit in asm blocks; the compiler's own are outside them)
        # (collected over the whole kernel: the first one precedes the first MFMA; the prologue's and the exit's own are vmcnt(0))
        waits, in_asm = [], False
        for ln in lines:
            t = ln.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
            elif t.startswith(";;#ASMEND"):
                in_asm = False
            elif in_asm and t.startswith("s_waitcnt vmcnt("):
                waits.append(int(re.search(r"vmcnt\((\d+)\)", t).group(1)))
        rh = 3 if (kt - 1) % 3 else 4
        rc = 1 if ntt == 6 else 2
        assert [w for w in waits if w > 0] == _f8_ages(kt, rh, rc, raw), (name, waits, _f8_ages(kt, rh, rc, raw))
        nb, ng = 2 * ntt, (kt + 1) // 4
        n16 = sum(1 for ln in body if "v_mfma_f32_16x16x32_f16" in ln)
        n8 = sum(1 for ln in body if "v_mfma_scale_f32_16x16x128_f8f6f4" in ln)
        assert (n16, n8) == (2 * kt * nb, 2 * 2 * ng * nb), (name, n16, n8)
        valu = sum(1 for ln in body if re.match(r"\s*v_(?!mfma)", ln))
        # vector instructions per matrix-pipe slot of 16 cycles (a scaled MFMA is two): the loop is bound by the SIMD's vector issue
        # as much as by the matrix pipe, so this must not quietly grow
        assert valu / (n16 + 2 * n8) <= {11: 2.0, 7: 3.0, 3: 6.5}[kt], (name, valu / (n16 + 2 * n8))


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_direct_a_gemm_assembly(tmp_path):
    ks = _kernels(_asm("conv_f16x3_dag.hip", tmp_path))
    assert ks, "no kernel found"
    for name, lines in ks.items():
        assert not any("scratch_" in ln for ln in lines), f"{name} spills"
        bad = _audit_no_touch_before_wait(lines)
        assert not bad, f"{name}: ring registers touched before their wait: {bad[:3]}"
        if "dagn_kernel" in name:
            # the narrow form, ring of R = 6 slots: every wait of the loop is vmcnt(10 (R - 2)) = "the R - 2 younger slots", so exactly
            # ten vector loads (2 weight fragments + 8 input values) must lie between two consecutive waits, and nothing else may
            # load or store; the one wait behind the ring's first fill is vmcnt(10 (R - 1))
            body = [ln.strip() for ln in lines]
            R = 6
            assert sum(1 for ln in body if ln.startswith(f"s_waitcnt vmcnt({10 * (R - 1)})")) == 1, f"{name}: the wait behind the first fill"
            waits = [i for i, ln in enumerate(body) if ln.startswith(f"s_waitcnt vmcnt({10 * (R - 2)})")]
            assert len(waits) == R, f"{name}: {len(waits)} ring waits in the unrolled round of {R}"
            for a0, a1 in zip(waits, waits[1:]):
                n = sum(1 for ln in body[a0:a1] if re.match(r"(global|buffer|flat|scratch)_(load|store|atomic)", ln))
                assert n == 10, f"{name}: {n} vector-memory operations between two ring waits, the waits assume 10"
            continue
        sizes = _prefetch_batches(lines)
        assert sizes and all(sz == 24 for sz in sizes), f"{name}: input prefetch is {sizes} loads, raw_ops assumes 24"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_pre_split_kernels_assembly(tmp_path):
    """Round 5: the pre-split forms of the direct-A kernel (input staged from an image: five or six 16-byte buffer loads per lane
    and chunk, counted by the ring's waits as `raw_ops = NLD`) and the narrow form without staging (an eight-slot ring of four
    asm loads per step, every wait vmcnt(28)).  No spills, no ring register touched before its wait, the prefetch batches and the
    ring waits exactly as counted."""
    ks = _kernels(_asm("conv_f16x3_da_pre.hip", tmp_path))
    assert len(ks) == 4, sorted(ks)
    for name, lines in ks.items():
        _audit_spills(name, lines)
        bad = _audit_no_touch_before_wait(lines)
        assert not bad, f"{name}: ring registers touched before their wait: {bad[:3]}"
        body = [ln.strip() for ln in lines]
        # the image loads: buffer_load_dwordx4 only (no dword loads of an f32 input, no transform left in the kernel)
        mf = [i for i, ln in enumerate(body) if ln.startswith("v_mfma")]
        assert not any(re.match(r"global_load_dword\s", ln) for ln in body[mf[0]:mf[-1]]), f"{name}: scalar-dword input loads in the main loop of a pre-split form"
        mm = re.search(r"da_kernelILi(\d+)ELi(\d+)ELi(\d+)E", name)
        kt, ntt = int(mm.group(2)), int(mm.group(3))
        nld = 4 * (32 * ntt + (64 if kt > 0 else 128)) // 256
        # every run of image loads between two MFMAs / barriers is one prefetch batch of exactly NLD loads
        runs, n = [], 0
        for ln in body:
            if ln.startswith("buffer_load_dwordx4"):
                n += 1
            elif (ln.startswith("v_mfma") or ln == "s_barrier") and n:
                runs.append(n)
                n = 0
        assert runs and all(r == nld for r in runs), f"{name}: image prefetch batches {sorted(set(runs))}, raw_ops assumes {nld}"
        assert sum(1 for ln in body if ln.startswith("v_sin") or ln.startswith("v_rndne")) == 0, f"{name}: snake arithmetic in a pre-split form"
    ks = _kernels(_asm("conv_f16x3_dapn.hip", tmp_path))
    assert len(ks) == 1, sorted(ks)
    for name, lines in ks.items():
        assert not any("scratch_" in ln for ln in lines), f"{name} spills"
        bad = _audit_no_touch_before_wait(lines)
        assert not bad, f"{name}: ring registers touched before their wait: {bad[:3]}"
        body = [ln.strip() for ln in lines]
        R = 8
        waits = [i for i, ln in enumerate(body) if ln.startswith(f"s_waitcnt vmcnt({4 * (R - 1)})")]
        assert len(waits) == R, f"{name}: {len(waits)} ring waits in the unrolled round of {R}"
        for a0, a1 in zip(waits, waits[1:]):
            nmem = sum(1 for ln in body[a0:a1] if re.match(r"(global|buffer|flat|scratch)_(load|store|atomic)", ln))
            assert nmem == 4, f"{name}: {nmem} vector-memory operations between two ring waits, the waits assume 4"
            nmf = sum(1 for ln in body[a0:a1] if ln.startswith("v_mfma"))
            assert nmf == 3, f"{name}: {nmf} MFMAs per step"
