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


def _asm(src, tmp_path, *flags):
    out = tmp_path / (os.path.basename(src) + ".s")
    subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "--cuda-device-only", "-S", *flags,
                    os.path.join(ROOT, "kokorox_amd", "csrc", src), "-o", str(out)], check=True, capture_output=True)
    return out.read_text()


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
    """Between an asm `global_load_dwordx4 vDST` and the next hand-written wait no other instruction may name vDST."""
    pending, in_asm, bad = set(), False, []
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
        if in_asm:
            if s.startswith("global_load_dwordx4") and toks:
                pending |= _regs(toks[0])
            elif s.startswith("s_waitcnt"):
                pending = set()
            continue
        used = set().union(*[_regs(t) for t in toks]) if toks else set()
        if used & pending:
            bad.append(s)
    return bad


def _prefetch_batches(lines):
    """Sizes of the runs of scalar-dword vector loads that follow an s_barrier (the input prefetch of a chunk boundary)."""
    sizes, i = [], 0
    while i < len(lines):
        if lines[i].strip() == "s_barrier":
            n, j = 0, i + 1
            while j < len(lines) and j < i + 2000 and "v_mfma" not in lines[j]:
                if re.match(r"\s*global_load_dword\s", lines[j]):
                    n += 1
                j += 1
            # (only barriers of the main loop: MFMAs follow; the epilogue's barrier is followed by its own loads)
            if n >= 8 and j < len(lines) and "v_mfma" in lines[j]:
                sizes.append(n)
            i = j
        else:
            i += 1
    return sizes


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_direct_a_conv_assembly(tmp_path):
    ks = _kernels(_asm("conv_f16x3_da.hip", tmp_path, "-DKX_DA_AUDIT"))
    assert ks, "no kernel found"
    for name, lines in ks.items():
        assert not any("scratch_" in ln for ln in lines), f"{name} spills"
        bad = _audit_no_touch_before_wait(lines)
        assert not bad, f"{name}: ring registers touched before their wait: {bad[:3]}"
        m = re.search(r"da_kernelILi(\d+)ELi(\d+)ELi(\d+)E", name)
        act, kt, ntt = (int(x) for x in m.groups())
        hu = (32 * ntt + 128) // 128 * 8
        want = hu + 3 + (1 if act == 2 else 0)  # raw_ops with the three InstanceNorm parameter loads present
        sizes = _prefetch_batches(lines)
        assert sizes and all(sz == want for sz in sizes), f"{name}: input prefetch is {sizes} loads, raw_ops assumes {want}"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_direct_a_gemm_assembly(tmp_path):
    ks = _kernels(_asm("conv_f16x3_dag.hip", tmp_path))
    assert ks, "no kernel found"
    for name, lines in ks.items():
        assert not any("scratch_" in ln for ln in lines), f"{name} spills"
        bad = _audit_no_touch_before_wait(lines)
        assert not bad, f"{name}: ring registers touched before their wait: {bad[:3]}"
        sizes = _prefetch_batches(lines)
        assert sizes and all(sz == 24 for sz in sizes), f"{name}: input prefetch is {sizes} loads, raw_ops assumes 24"
