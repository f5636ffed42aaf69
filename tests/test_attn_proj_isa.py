"""Build-time checks on the compiled ISA of the one-workgroup-per-image kernels (csrc/attn_proj.hip, csrc/panel_gemm.hip), CPU only (hipcc cross-compiles
gfx950 without a GPU) — the two compiler facts these kernels were built around, as assertions:

1. hipcc answers a TRACKED vector-memory operation that crosses an LDS-DMA with `s_waitcnt vmcnt(0)` at its first use, draining the prefetch it was meant to
   run beside.  Inside the pipelined region (from the kernel's first workgroup barrier to its last counted wait) every wait on the vector-memory counter must
   therefore be one of OURS (inline asm), never the compiler's.
2. The compiler takes an inline asm's outputs as valid the moment the statement has executed: nothing may touch a register an asm load / LDS read has written
   before the hand-placed wait that covers it (it once copied the loop-carried Q registers at the loop latch, in front of the wait, and re-used their registers
   for addresses the late data then overwrote: a memory fault at >= 128 images only).  Followed along the control flow; vector-memory operations retire in
   issue order (loads, LDS-DMA and stores share one counter on gfx950), LDS reads in theirs.
Plus: no scratch."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

VMEM_LOAD = ("global_load_", "buffer_load_", "flat_load_", "scratch_load_")
VMEM_STORE = ("global_store_", "buffer_store_", "flat_store_", "global_atomic_", "buffer_atomic_")
LDS_READ = ("ds_read_", "ds_bpermute", "ds_permute", "ds_swizzle")


def _compile(tmp_path, src):
    out = tmp_path / (os.path.basename(src) + ".s")
    res = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wall", "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage", "-S",
                          "--cuda-device-only", "-o", str(out), src], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    return out.read_text(), res.stderr


def _regs(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def _kernels(isa):
    """[(name, [lines])] per .amdhsa kernel body (from its label to s_endpgm's function end marker)"""
    out, cur, name = [], None, None
    for ln in isa.splitlines():
        m = re.match(r"^(_Z\w+):", ln)
        if m and "attn_proj_kernel" in m.group(1) or m and "panel_gemm_kernel" in m.group(1):
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if ln.startswith(".Lfunc_end"):
                out.append((name, cur))
                cur = None
            else:
                cur.append(ln.strip())
    return out


def _walk(lines):
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

    n_asm_loads = [0]

    def run_block(name, state):
        """-> [(successor label, state there)]: a branch in the middle of a labelled run hands its target the state AT the branch"""
        vm, lds = [set(x) for x in state[0]], [set(x) for x in state[1]]     # outstanding ops, oldest first: the registers they will write (maybe none)
        in_asm = False
        edges = []
        snap = lambda: (tuple(frozenset(x) for x in vm), tuple(frozenset(x) for x in lds))
        for ln in blocks[name]:
            op0, _, rest0 = ln.partition(" ")
            if op0.startswith("s_cbranch") or op0 == "s_branch":
                edges.append((rest0.strip(), snap()))
                if op0 == "s_branch":
                    return edges
                continue
            if op0 in ("s_endpgm", "s_setpc_b64"):
                return edges
            if ln.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if ln.startswith(";;#ASMEND"):
                in_asm = False
                continue
            op, _, rest = ln.partition(" ")
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", ln)
                if m:
                    keep = int(m.group(1))
                    vm = vm[len(vm) - keep:] if keep else []
                m = re.search(r"lgkmcnt\((\d+)\)", ln)
                if m:
                    keep = int(m.group(1))
                    lds = lds[len(lds) - keep:] if keep else []
                continue
            pending = (set().union(*vm) if vm else set()) | (set().union(*lds) if lds else set())
            touched = _regs(rest) & pending
            if op.startswith(VMEM_LOAD):
                dst, _, addr = rest.partition(",")
                is_dma = " lds" in (" " + rest) or op.startswith("global_load_lds")
                assert not (_regs(addr) & pending) and (is_dma or not (_regs(dst) & pending)), f"{name}: {ln!r} uses a register whose load has not been waited for"
                # only an ASM load's destination is at risk (the compiler waits for its own); a tracked load still occupies a counter slot
                vm.append(set() if (is_dma or not in_asm) else _regs(dst))
                n_asm_loads[0] += int(in_asm and not is_dma)
                continue
            if op.startswith(VMEM_STORE):
                assert not touched, f"{name}: {ln!r} stores a register whose load has not been waited for"
                vm.append(set())
                continue
            if op.startswith(LDS_READ):
                dst, _, addr = rest.partition(",")
                assert not (_regs(addr) & pending), f"{name}: {ln!r} addresses through an un-waited register"
                lds.append(_regs(dst) if in_asm else set())
                continue
            if op.startswith("ds_"):     # LDS writes / others: counted, no destination
                assert not touched, f"{name}: {ln!r} touches v{sorted(touched)} before the wait that covers its load"
                lds.append(set())
                continue
            if op.startswith(("s_load", "s_buffer_load")):
                lds.append(set())        # (scalar loads share lgkmcnt; the compiler waits lgkmcnt(0) for them: retired there)
                continue
            assert not touched, f"{name}: {ln!r} touches v{sorted(touched)} before the wait that covers its load"
        i = order.index(name)
        if i + 1 < len(order):
            edges.append((order[i + 1], snap()))
        return edges

    state_in = {"<entry>": ((), ())}
    work = ["<entry>"]
    guard = 0
    while work:
        guard += 1
        assert guard < 20000, "the control-flow walk does not converge"
        name = work.pop()
        for nxt, out in run_block(name, state_in[name]):
            if nxt not in blocks:
                continue
            old = state_in.get(nxt)
            if old is None:
                merged = out
            else:
                # a join with different histories: per counter, the longer history's length with the union of pending registers in its oldest group
                merged = []
                for o, n in zip(old, out):
                    if o == n:
                        merged.append(o)
                    else:
                        keep = max(len(o), len(n))
                        regs = frozenset().union(*o, *n) if (o or n) else frozenset()
                        merged.append(tuple([regs] + [frozenset()] * (keep - 1)) if keep else ())
                merged = tuple(merged)
            if merged != old:
                state_in[nxt] = merged
                work.append(nxt)
    return n_asm_loads[0]


def _check_kernel(name, lines, whole_region):
    body = [ln for ln in lines if ln and (not ln.startswith((";", ".", "//")) or ln.startswith(";;#ASM"))]
    # 1. every vmcnt wait of the pipeline is an asm statement of ours.  attn_proj: from the kernel's first workgroup barrier to its last counted wait (its
    # epilogue lies behind).  panel_gemm: inside the straight-line 12-unit GEMM body (96 MFMAs between two labels) — a pass's epilogue, between two runs of
    # that body, uses tracked loads on purpose (bias / colsum: the compiler waits for them, and with them for the three units in flight, once per pass)
    if whole_region:
        barriers = [i for i, ln in enumerate(body) if ln.startswith("s_barrier")]
        counted = [i for i, ln in enumerate(body) if ln.startswith("s_waitcnt") and re.search(r"vmcnt\((6|18)\)", ln)]
        assert barriers and counted, name
        spans = [(barriers[0], counted[-1])]
    else:
        spans, start, n = [], 0, 0
        for i, ln in enumerate(lines + [".LBBend_0:"]):
            if re.match(r"^\.LBB\w+:", ln):
                if n >= 96:
                    spans.append((start, i))
                start, n = i, 0
            elif ln.startswith("v_mfma_f32_16x16x32_bf16"):
                n += 1
        assert spans, name
        body = lines
    for lo, hi in spans:
        for i in range(lo, min(hi + 1, len(body))):
            if body[i].startswith("s_waitcnt") and "vmcnt" in body[i]:
                prev = next(body[j] for j in range(i - 1, -1, -1) if body[j])
                assert prev.startswith(";;#ASMSTART"), f"{name}: the compiler waits on the vector-memory counter inside the pipeline: {body[i]!r} (line {i})"
    # 2. no register of an asm load / read is touched before its wait
    return _walk(lines)


@pytest.mark.skipif(not shutil.which(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("unit,min_kernels", [("attn_proj", 1), ("panel_gemm", 3)])
def test_no_compiler_wait_in_the_pipeline_and_no_use_before_the_wait(tmp_path, unit, min_kernels):
    isa, remarks = _compile(tmp_path, os.path.join(ROOT, "marqo_amd", "csrc", unit + ".hip"))
    assert not re.search(r"ScratchSize \[bytes/lane\]: [1-9]", remarks) and not re.search(r"VGPRs Spill: [1-9]", remarks), remarks[-1500:]
    kernels = _kernels(isa)
    assert len(kernels) >= min_kernels, [k for k, _ in kernels]
    for name, lines in kernels:
        n_mfma = sum(1 for ln in lines if ln.startswith("v_mfma_f32_16x16x32_bf16"))
        assert n_mfma >= 96, (name, n_mfma)            # the 12-unit GEMM body (96) [+ the attention rounds]
        n_asm_loads = _check_kernel(name, lines, whole_region=(unit == "attn_proj"))
        if unit == "attn_proj":
            assert n_asm_loads >= 2 + 6 * 2            # the Q rows of every round at least (+ residual quads, weight-line touches)
