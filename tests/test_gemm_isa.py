"""Build-time checks on the compiled ISA of the bf16 GEMM main loop (csrc/gemm_bf16.hip), CPU only (hipcc cross-compiles gfx950 without a GPU).

The k-loop issues its LDS fragment reads as inline asm (hipcc cannot tell them from the LDS-DMA writes in flight and would serialise the pipeline with
`s_waitcnt vmcnt(0)` in front of every read) and places the waits by hand.  The compiler takes an asm's outputs as valid the moment the statement has
executed, so nothing may touch a fragment register between its `ds_read_b128` and the hand-placed `s_waitcnt lgkmcnt(0)`: a register copy the allocator
drops in there would read stale bits and only fail on some data, some day.  This test reads the ISA and checks exactly that, plus: no scratch, ONE
straight-line k-step body (8 * MT MFMAs in the whole kernel), ONE vmcnt wait inside it (the mid-step one — a second one means the compiler started
waiting on the LDS-DMA behind our back)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "marqo_amd", "csrc", "gemm_bf16.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

BIAS, GELU, RESIDUAL, OUT_F32, ROW_STATS, LN_APPLY, GLU = 1, 2, 8, 16, 64, 128, 256


def _compile(tmp_path, flags, mt, src=SRC, rowscale=0, nh=1, wm=2):
    out = tmp_path / f"probe_{os.path.basename(src)}_{flags}_{mt}_{rowscale}_{nh}_{wm}.s"
    res = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wall", "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage",
                          f"-DMQ_GEMM_PROBE={flags}", f"-DMQ_GEMM_PROBE_MT={mt}", f"-DMQ_GEMM_PROBE_ROWSCALE={rowscale}", f"-DMQ_GEMM_PROBE_NH={nh}", f"-DMQ_GEMM_PROBE_WM={wm}", "-S", "--cuda-device-only", "-o", str(out), src],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    return out.read_text(), res.stderr


def _regs(operand_text):
    """VGPR indices named in an operand string: v12, v[12:15]"""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", operand_text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", operand_text):
        out.add(int(m.group(1)))
    return out


def _check(isa, remarks, mfma_name, n_mfma, n_reads_expected, hot_regions=1):
    assert re.search(r"ScratchSize \[bytes/lane\]: 0\b", remarks) and re.search(r"VGPRs Spill: 0\b", remarks), remarks[-1500:]
    lines = [ln.strip() for ln in isa.splitlines()]
    body = [ln for ln in lines if ln and not ln.startswith((";", ".", "//")) or ln.startswith(";;#ASM")]
    # 1. straight-line k-steps: the expected number of MFMAs, nowhere else
    mfma = [i for i, ln in enumerate(body) if ln.startswith(mfma_name)]
    assert len(mfma) == n_mfma, len(mfma)
    # 2. per k-step body exactly one wait on the vector-memory counter (the mid-step one: a second one means the compiler started waiting on the
    # LDS-DMA behind our back) and one barrier
    per = n_mfma // hot_regions
    for r in range(hot_regions):
        hot = body[mfma[r * per]:mfma[(r + 1) * per - 1] + 1]
        assert sum(1 for ln in hot if ln.startswith("s_waitcnt") and "vmcnt" in ln) == 1, [ln for ln in hot if "vmcnt" in ln]
        assert sum(1 for ln in hot if ln.startswith("s_barrier")) == 1
    # 3. no instruction touches a register an inline-asm ds_read has written before the next lgkmcnt wait that covers it — followed along the
    # control flow (the compiler lays blocks out of line: behind an unconditional branch the textually next instruction is NOT the next one executed)
    # (LDS operations return in order: `s_waitcnt lgkmcnt(N)` retires all but the N youngest reads)
    label_re = re.compile(r"^(\.LBB\d+_\d+):")
    blocks, order, cur = {"<entry>": []}, ["<entry>"], "<entry>"
    for ln in lines:
        m = label_re.match(ln)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            order.append(cur)
            continue
        if ln and (not ln.startswith((";", ".", "//")) or ln.startswith(";;#ASM")):
            blocks[cur].append(ln)

    def successors(name):
        out, falls = [], True
        for ln in blocks[name]:
            op, _, rest = ln.partition(" ")
            if op.startswith("s_cbranch"):
                out.append(rest.strip())
            elif op == "s_branch":
                out.append(rest.strip())
                falls = False
            elif op in ("s_endpgm", "s_setpc_b64"):
                falls = False
        i = order.index(name)
        if falls and i + 1 < len(order):
            out.append(order[i + 1])
        return out

    counted = set()

    def run_block(name, pending_in):
        """pending_in: tuple of frozensets (un-retired asm reads, oldest first) -> state at the end of the block"""
        reads, in_asm = list(pending_in), False
        for k, ln in enumerate(blocks[name]):
            if ln.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if ln.startswith(";;#ASMEND"):
                in_asm = False
                continue
            m = re.search(r"lgkmcnt\((\d+)\)", ln) if ln.startswith("s_waitcnt") else None
            if m:
                keep = int(m.group(1))
                reads = reads[len(reads) - keep:] if keep else []
                continue
            pending = set().union(*reads) if reads else set()
            op, _, rest = ln.partition(" ")
            if in_asm and op == "ds_read_b128":
                dst, _, addr = rest.partition(",")
                assert not (_regs(addr) & pending), f"asm read addresses through an un-waited register: {ln}"
                reads.append(frozenset(_regs(dst)))
                counted.add((name, k))
                continue
            touched = _regs(rest) & pending
            assert not touched, f"{name}: {ln!r} touches v{sorted(touched)} before the s_waitcnt lgkmcnt that covers its ds_read_b128"
        return tuple(reads)

    state_in = {"<entry>": ()}
    work = ["<entry>"]
    while work:
        name = work.pop()
        out = run_block(name, state_in[name])
        for nxt in successors(name):
            if nxt not in blocks:
                continue
            old = state_in.get(nxt)
            if old is None:
                merged = out
            elif old == out or not out:
                merged = old
            else:   # a join with different histories: everything pending on either path, as ONE oldest group (conservative)
                merged = (frozenset().union(*old, *out),)
            if merged != old:
                state_in[nxt] = merged
                work.append(nxt)
    n_reads = len(counted)
    assert n_reads == n_reads_expected, n_reads


@pytest.mark.skipif(not shutil.which(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("flags,mt", [(BIAS, 5), (BIAS | RESIDUAL, 5), (BIAS | GELU, 4), (BIAS | RESIDUAL | OUT_F32, 6), (BIAS | GELU | LN_APPLY, 5), (BIAS | RESIDUAL | ROW_STATS, 5),
                                      (BIAS | RESIDUAL | ROW_STATS, 3), (0, 2), (BIAS | GLU | LN_APPLY, 5), (BIAS | GLU, 6),
                                      (BIAS | RESIDUAL | ROW_STATS | LN_APPLY, 5), (BIAS | RESIDUAL | ROW_STATS | LN_APPLY, 4), (BIAS | GLU | LN_APPLY | ROW_STATS, 6)])
def test_asm_fragment_reads_are_waited_for_before_any_use(tmp_path, flags, mt):
    isa, remarks = _compile(tmp_path, flags, mt)
    _check(isa, remarks, "v_mfma_f32_16x16x32_bf16", 8 * mt, 3 * (mt + 4))    # reads: prologue + both half-steps


@pytest.mark.skipif(not shutil.which(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("flags", [0, BIAS, BIAS | GELU, BIAS | RESIDUAL, BIAS | RESIDUAL | OUT_F32, BIAS | GELU | LN_APPLY, BIAS | RESIDUAL | ROW_STATS,
                                   BIAS | RESIDUAL | ROW_STATS | LN_APPLY, BIAS | GLU | LN_APPLY | ROW_STATS])
def test_big_8_wave_tile_follows_the_same_rules(tmp_path, flags):
    """round 5: the 256 x 256 tile of 8 waves (WM = 4, MT = 4, NH = 2: 64 x 128 per wave, two waves per SIMD, 8 LDS-DMA pieces and 12 fragment reads per
    wave and k-half): the same loop — 2 * 32 MFMAs in ONE k-step body, one vmcnt wait, one barrier, no scratch inside the 256 registers of a
    2-waves-per-SIMD lane; fragment reads: prologue + both half-steps + the re-read behind the epilogue = 4 * 12"""
    isa, remarks = _compile(tmp_path, flags, 4, nh=2, wm=4)
    # two k-step bodies: the tile walk's and the in-kernel tail's (the ragged last row of tiles, cut along K over the grid); fragment reads:
    # (prologue + both half-steps) x 2 phases + the re-read behind the tile walk's epilogue
    _check(isa, remarks, "v_mfma_f32_16x16x32_bf16", 2 * 2 * 32, 7 * 12, hot_regions=2)
    assert re.search(r"Occupancy \[waves/SIMD\]: 2\b", remarks)


OUT_FP8 = 32


@pytest.mark.skipif(not shutil.which(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("flags,mt,rowscale", [(BIAS, 5, 1), (BIAS | GELU | OUT_FP8, 6, 1), (BIAS | RESIDUAL, 6, 0), (BIAS | RESIDUAL | OUT_F32, 5, 0), (OUT_F32, 4, 0),
                                               (BIAS | GELU | OUT_FP8, 2, 1)])
def test_fp8_loop_follows_the_same_rules(tmp_path, flags, mt, rowscale):
    """csrc/gemm_fp8.hip: the same pipeline (a k-step = two W-side halves; a tile's last k-step is a second body without the next stage's fragment
    reads, which the tile loop issues after the epilogue): two k-step bodies of 4 * MT MFMAs each; fragment reads (two ds_read_b128 each): MT + 2 at
    the tile top, MT + 4 in the steady k-step, 2 in the last one"""
    isa, remarks = _compile(tmp_path, flags, mt, src=SRC.replace("gemm_bf16.hip", "gemm_fp8.hip"), rowscale=rowscale)
    _check(isa, remarks, "v_mfma_scale_f32_16x16x128_f8f6f4", 8 * mt, 2 * (mt + 2) + 2 * (mt + 4) + 2 * 2, hot_regions=2)


@pytest.mark.skipif(not shutil.which(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("flags,rowscale", [(BIAS, 1), (BIAS | GELU | OUT_FP8, 1), (BIAS | RESIDUAL, 0), (BIAS | RESIDUAL | OUT_F32, 0), (OUT_F32, 0)])
def test_fp8_big_tile_follows_the_same_rules(tmp_path, flags, rowscale):
    """round 6: the 8-wave 192 x 256 x 128 tile of csrc/gemm_fp8.hip (WM = 4, NH = 2, MT = 3: 48 x 128 per wave, 8 W sub-tiles, two waves per SIMD): the
    same two k-step bodies — 8 * MT MFMAs each — one vmcnt wait and one barrier per body, no scratch inside 256 registers; fragment reads (two
    ds_read_b128 each): MT + 4 at the tile top, MT + 8 in the steady k-step, 4 in a tile's last one"""
    mt = 3
    isa, remarks = _compile(tmp_path, flags, mt, src=SRC.replace("gemm_bf16.hip", "gemm_fp8.hip"), rowscale=rowscale, nh=2, wm=4)
    _check(isa, remarks, "v_mfma_scale_f32_16x16x128_f8f6f4", 2 * 8 * mt, 2 * (mt + 4) + 2 * (mt + 8) + 2 * 4, hot_regions=2)
    assert re.search(r"Occupancy \[waves/SIMD\]: 2\b", remarks)
