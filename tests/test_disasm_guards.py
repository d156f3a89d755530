"""Deterministic checks of the EMITTED gfx950 code (no GPU): the similarity tile kernel's k-loop depends on three things the compiler
is free to break and a parity run only catches by luck (csrc/coarse_match.hip ``sim_frag_kernel``; round 3 saw "rare wrong tiles" when hipcc
moved LDS reads across an ``s_barrier``):

* every k-step opens with ONE statement ``s_waitcnt vmcnt(N) lgkmcnt(0)`` + ``s_barrier`` (nothing between the two);
* the counted wait is only right if exactly the LDS-DMA pieces of chunks 0 .. s + 2 have been issued when step s waits (``vmcnt(2 G)``
  then leaves chunks s + 1 and s + 2 in flight and nothing of chunk s) -- an issue that slips behind the wait makes it pass too early, one
  that is hoisted in front of the previous barrier overwrites a buffer that is still being read;
* the fragments of step s are read from ring buffer s % 4 and nowhere else, between barrier s and barrier s + 1.

The test disassembles the code object inside the built ``libonepose_hip.so`` (``llvm-objdump`` of the ROCm install) and asserts all three
for every instantiation of the kernel.  Reference for what the kernel computes: ``utils/coarse_matching.py:101-115``."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

from onepose_st_amd import hip

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
NKS = 16                       # k-steps of the 256-feature contraction (16 features each)
CHUNK = 16384                  # bytes of one ring buffer (FRAG_CHUNK_BYTES)


@pytest.fixture(scope="module")
def device_asm():
    """{kernel symbol: [instruction lines]} of every gfx950 code object bundled in the library"""
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump of the ROCm install not found")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(hip.library_path(), so)
        subprocess.run([OBJDUMP, "--offloading", so], cwd=tmp, check=True, capture_output=True)          # writes lib.so.<n>.<target> files
        for name in sorted(os.listdir(tmp)):
            if not name.endswith("gfx950"):
                continue
            text = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, name)], check=True, capture_output=True, text=True).stdout
            cur = None
            for line in text.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
                if m:
                    cur = out.setdefault(m.group(1), [])
                elif cur is not None and line.startswith("\t"):
                    cur.append(line.strip().split("//")[0].strip())
    return out


def _sim_frag_kernels(device_asm):
    ks = {k: v for k, v in device_asm.items() if "sim_frag_kernel" in k}
    assert len(ks) == 8, sorted(ks)                                   # NS in {1, 3} x MODE in {0, 1, 2, 3}
    return ks


def test_sim_frag_k_loop_keeps_its_wait_barrier_and_ring_discipline(device_asm):
    for sym, ins in _sim_frag_kernels(device_asm).items():
        ns = 3 if "ILi3E" in sym else 1
        G = 4 if ns == 3 else 2                                       # LDS-DMA pieces per chunk and wave (A + B, hi + lo)
        # the k-loop's sixteen wait + barrier statements: a counted vmcnt WITH lgkmcnt(0), directly followed by s_barrier
        opens = [i for i, t in enumerate(ins) if re.match(r"s_waitcnt vmcnt\(\d+\) lgkmcnt\(0\)$", t)]
        opens = [i for i in opens if i + 1 < len(ins) and ins[i + 1] == "s_barrier"]
        assert len(opens) >= NKS, (sym, len(opens))
        opens = opens[:NKS]
        counts = [int(re.search(r"vmcnt\((\d+)\)", ins[i]).group(1)) for i in opens]
        assert counts == [2 * G] * (NKS - 2) + [G, 0], (sym, counts)
        # no other vector-memory instruction inside the loop: vmcnt counts loads, stores and LDS-DMA alike, in issue order
        first_dma = next(i for i, t in enumerate(ins) if t.startswith("global_load_lds_dwordx4"))
        for t in ins[first_dma:opens[-1]]:
            if re.match(r"(global|buffer|scratch|flat)_", t):
                assert t.startswith("global_load_lds_dwordx4"), (sym, t)
        dma_before = lambda idx: sum(1 for t in ins[:idx] if t.startswith("global_load_lds_dwordx4"))
        for s, at in enumerate(opens):
            # chunks 0 .. min(s + 2, NKS - 1) issued, no more (chunk s + 3 goes into the buffer read in step s - 1) and no fewer (the count)
            assert dma_before(at) == G * min(s + 3, NKS), (sym, s, dma_before(at))
            # fragment reads of this step: ring buffer s % 4, between this barrier and the next
            end = opens[s + 1] if s + 1 < NKS else next(i for i in range(at + 2, len(ins)) if ins[i] == "s_barrier")
            reads = [t for t in ins[at + 2:end] if t.startswith("ds_read")]
            assert len(reads) == 2 * G, (sym, s, reads)
            for t in reads:
                assert t.startswith("ds_read_b128"), (sym, t)
                off = re.search(r"offset:(\d+)", t)
                off = int(off.group(1)) if off else 0
                assert off // CHUNK == s % 4, (sym, s, t)
        # nothing reads the ring before the first barrier
        assert not any(t.startswith("ds_read") for t in ins[:opens[0]]), sym


def test_sim_frag3_k_loop_keeps_its_wait_barrier_and_ring_discipline(device_asm):
    """The three-buffer tile kernel (sim_frag3_kernel, OPHIP_SIM_TILE=3): chunks run TWO k-steps ahead, so step s waits with
    vmcnt(G) (chunk s + 1 may stay in flight; vmcnt(0) in the last step), exactly the pieces of chunks 0 .. s + 1 have been issued when it
    waits, chunk s + 2 is issued behind the barrier into buffer (s + 2) % 3 = the one read in step s - 1, and step s reads buffer s % 3 only."""
    ks = {k: v for k, v in device_asm.items() if "sim_frag3_kernel" in k}
    assert len(ks) == 4, sorted(ks)                                   # NS in {1, 3} x MODE in {0, 1}
    for sym, ins in ks.items():
        ns = 3 if "ILi3ELi" in sym else 1
        G = 4 if ns == 3 else 2
        opens = [i for i, t in enumerate(ins) if re.match(r"s_waitcnt vmcnt\(\d+\) lgkmcnt\(0\)$", t)]
        opens = [i for i in opens if i + 1 < len(ins) and ins[i + 1] == "s_barrier"]
        assert len(opens) >= NKS, (sym, len(opens))
        opens = opens[:NKS]
        counts = [int(re.search(r"vmcnt\((\d+)\)", ins[i]).group(1)) for i in opens]
        assert counts == [G] * (NKS - 1) + [0], (sym, counts)
        first_dma = next(i for i, t in enumerate(ins) if t.startswith("global_load_lds_dwordx4"))
        for t in ins[first_dma:opens[-1]]:
            if re.match(r"(global|buffer|scratch|flat)_", t):
                assert t.startswith("global_load_lds_dwordx4"), (sym, t)
        dma_before = lambda idx: sum(1 for t in ins[:idx] if t.startswith("global_load_lds_dwordx4"))
        for s_, at in enumerate(opens):
            assert dma_before(at) == G * min(s_ + 2, NKS), (sym, s_, dma_before(at))
            end = opens[s_ + 1] if s_ + 1 < NKS else next(i for i in range(at + 2, len(ins)) if ins[i] == "s_barrier")
            reads = [t for t in ins[at + 2:end] if t.startswith("ds_read")]
            assert len(reads) == 2 * G, (sym, s_, reads)
            for t in reads:
                assert t.startswith("ds_read_b128"), (sym, t)
                off = re.search(r"offset:(\d+)", t)
                off = int(off.group(1)) if off else 0
                assert off // CHUNK == s_ % 3, (sym, s_, t)
        assert not any(t.startswith("ds_read") for t in ins[:opens[0]]), sym


def test_fine_stage_reads_the_3d_token_once_per_register(device_asm):
    """The fine stage's last step broadcasts the 3D token's features with v_readlane (lanes 25 and 57) -- one pair per residual register.
    hipcc 7.2 reads ELEMENT 0 when a bit_cast is applied to a vector element directly (seen in round 5: one pair for sixteen registers,
    every test of the whole path red); the kernels copy the element first, and this test counts the instructions."""
    for part, tiles in (("fine_pair_kernelILi3E", 2), ("fine_pair_kernelILi1E", 2), ("fine_refine_bf16_kernelILi3ELi1E", 1), ("fine_refine_bf16_kernelILi1ELi1E", 1)):
        hits = [k for k in device_asm if part in k]
        assert len(hits) == 1, (part, hits)
        ins = device_asm[hits[0]]
        for ln in (25, 57):
            n = sum(1 for t in ins if re.match(rf"v_readlane_b32 s\d+, v\d+, {ln}$", t))
            assert n == 16 * tiles, (part, ln, n)


def test_hot_kernels_do_not_spill_beyond_what_is_documented(device_asm):
    """scratch traffic in a matrix kernel is a silent 2-5x: the encoder and similarity kernels must have none; the fine pair kernel's
    few spilled registers are bounded."""
    def scratch_ops(sym_part):
        hits = [k for k in device_asm if sym_part in k]
        assert hits, sym_part
        return {k: sum(1 for t in device_asm[k] if t.startswith("scratch_")) for k in hits}
    for part in ("enc_x3w8_kernel", "sim_frag_kernel", "sim_frag3_kernel", "conf_kernel"):
        assert all(v == 0 for v in scratch_ops(part).values()), scratch_ops(part)
    for k, v in scratch_ops("fine_pair_kernelILi3E").items():
        assert v <= 24, (k, v)
