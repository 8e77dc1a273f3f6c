#!/usr/bin/env python3
"""Static check of the hand-ordered vector-memory code in csrc/ (ADVICE r03).

Three kernels families issue vector-memory requests from inline asm and wait for them with COUNTED `s_waitcnt vmcnt(N)`
literals (the compiler's own bookkeeping drained vmcnt(0) in front of every use):
  mt19937.hip      mt_jump_mfma_kernel      A-operand ring: global_load_dwordx4 PF steps ahead, vmcnt(PF - 1)
  hamming_mfma.hip hamming_mfma_kernel<*>   LDS ring filled by global_load_lds_dwordx4, vmcnt(PPW * younger entries)
  dense_mfma.hip   gemm_dma_kernel<*>       LDS-DMA staging
Those literals are right only while (1) no other vector-memory instruction -- i.e. no register spill -- sits between a request
and its wait, and (2) nothing reads or writes a load's destination VGPRs before the wait that covers it (r03's memory fault:
the ring's last loads were still in flight when the compiler reused their registers for store addresses).  Both are properties
of the generated ISA, so they are checked on the ISA, not by a race test:

  * per kernel, from the code object's metadata: no scratch, no spilled VGPRs;
  * a linear simulation of the vmcnt FIFO over every kernel that contains inline-asm vector memory: every vector-memory
    instruction enters the FIFO, every `s_waitcnt vmcnt(N)` retires all but the N youngest entries, loop bodies are walked
    twice (state carried over the back edge), and NO instruction may name a VGPR that is the destination of an entry still in
    the FIFO;
  * between the first and the last inline-asm vector-memory request / wait of such a kernel (text order) every vector-memory
    instruction must come from an inline-asm block (a compiler-generated one would shift the counts).

usage: check_asm_contracts.py file.s [file.s ...]      (files from `hipcc -S --cuda-device-only`; exit status 1 on a violation)
"""
from __future__ import annotations

import re
import sys

VMEM_PREFIX = ("global_", "buffer_", "flat_", "scratch_", "tbuffer_")
RE_VRANGE = re.compile(r"\bv\[(\d+):(\d+)\]")
RE_VSINGLE = re.compile(r"\bv(\d+)\b")
RE_LABEL = re.compile(r"^(\.LBB[0-9_]+):")
RE_KERNEL = re.compile(r"^(_Z[A-Za-z0-9_]+):")
RE_VMCNT = re.compile(r"vmcnt\((\d+)\)")


class Violation(Exception):
    pass


def vgprs(text):
    regs = set()
    for a, b in RE_VRANGE.findall(text):
        regs.update(range(int(a), int(b) + 1))
    for a in RE_VSINGLE.findall(RE_VRANGE.sub(" ", text)):
        regs.add(int(a))
    return regs


def split_kernels(asm):
    """-> {name: [(lineno, text, in_asm_block)]} for every kernel function of the file"""
    kernels, cur, name, in_asm = {}, None, None, False
    for no, raw in enumerate(asm.splitlines(), 1):
        m = RE_KERNEL.match(raw)
        if m and cur is None:
            name, cur, in_asm = m.group(1), [], False
            continue
        if cur is None:
            continue
        line = raw.split(";")[0].rstrip() if not raw.lstrip().startswith(";;#") else raw.strip()
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        s = line.strip()
        if not s or s.startswith(";") or (s.startswith(".") and not RE_LABEL.match(s)):
            if s.startswith(".Lfunc_end"):
                kernels[name] = cur
                cur = None
            continue
        cur.append((no, s, in_asm))
    return kernels


def kernel_metadata(asm):
    """per kernel: scratch bytes, spill counts, VGPRs -- the `amdhsa.kernels:` list of the .amdgpu_metadata YAML, entry by entry"""
    out = {}
    entries = re.split(r"\n\s*- \.agpr_count:", asm)
    for e in entries[1:]:
        body = ".agpr_count:" + e
        d = {}
        for k in ("name", "private_segment_fixed_size", "sgpr_spill_count", "vgpr_spill_count", "vgpr_count", "agpr_count"):
            m = re.search(r"\." + k + r":\s*(\S+)", body)
            if m:
                d[k] = m.group(1) if k == "name" else int(m.group(1))
        if "name" in d:
            out[d["name"]] = d
    return out


def is_vmem(mn):
    return mn.startswith(VMEM_PREFIX)


def check_kernel(name, insts, max_states=200000):
    """Explores the kernel's control-flow graph with the vmcnt FIFO as the state (every distinct FIFO that can reach a basic
    block is walked through it once); returns a list of messages (empty = fine).
    FIFO entry = (destination VGPRs, issued from inline asm?, line).  The hardware counter holds at most 64 requests."""
    if not any(a and is_vmem(t.split()[0]) for _, t, a in insts):
        return []
    # ---- basic blocks ----
    leaders = {0}
    labels = {}
    for i, (_, t, _) in enumerate(insts):
        m = RE_LABEL.match(t)
        if m:
            labels[m.group(1)] = i
            leaders.add(i)
        if re.match(r"s_c?branch|s_endpgm|s_setpc", t) and i + 1 < len(insts):
            leaders.add(i + 1)
    starts = sorted(leaders)
    block_of = {}
    blocks = []
    for b, st in enumerate(starts):
        en = starts[b + 1] if b + 1 < len(starts) else len(insts)
        blocks.append((st, en))
        block_of[st] = b

    def successors(b):
        st, en = blocks[b]
        t = insts[en - 1][1]
        m = re.match(r"s_branch\s+(\S+)", t)
        if m:
            return [block_of[labels[m.group(1)]]] if m.group(1) in labels else []
        if t.startswith("s_endpgm") or t.startswith("s_setpc"):
            return []
        out = []
        m = re.match(r"s_cbranch_\w+\s+(\S+)", t)
        if m and m.group(1) in labels:
            out.append(block_of[labels[m.group(1)]])
        if en < len(insts):
            out.append(block_of[en])
        return out

    msgs, seen_msgs = [], set()

    def say(m):
        if m not in seen_msgs:
            seen_msgs.add(m)
            msgs.append(m)

    # every instruction parsed once: (kind, VGPRs named, destination VGPRs of a load, vmcnt literal)
    parsed = []
    for no, t, a in insts:
        if RE_LABEL.match(t):
            parsed.append(None)
            continue
        mn = t.split()[0]
        used = frozenset(vgprs(t[len(mn):]))
        if is_vmem(mn):
            dest = frozenset()
            if "atomic" in mn:
                say(f"{name}: line {no}: atomic in a kernel with counted vmcnt waits: not modelled")
            elif "_load" in mn and "_lds_" not in mn:
                dest = frozenset(vgprs(t[len(mn):].split(",")[0]))
            parsed.append(("vmem", used, dest, None))
        elif mn == "s_waitcnt":
            m = RE_VMCNT.search(t)
            if m:
                parsed.append(("wait", used, None, int(m.group(1))))
            elif not re.search(r"lgkmcnt|expcnt", t):
                parsed.append(("wait", used, None, 0))           # raw immediate: everything waited
            else:
                parsed.append(("other", used, None, None))
        else:
            parsed.append(("other", used, None, None))

    def run_block(b, fifo):
        fifo = list(fifo)
        st, en = blocks[b]
        for i in range(st, en):
            p = parsed[i]
            if p is None:
                continue
            kind, used, dest, n = p
            no, t, a = insts[i]
            if used and fifo:
                for fd, _, lno, ltxt in fifo:
                    if fd and (fd & used):
                        say(f"{name}: line {no}: `{t}` names v{sorted(fd & used)} while the load of line {lno} (`{ltxt}`) into "
                            f"them may still be in flight (no s_waitcnt vmcnt covers it on some path)")
            if kind == "vmem":
                fifo.append((dest, a, no, t))
                if len(fifo) > 64:
                    del fifo[0]
            elif kind == "wait":
                if a and n > 0:
                    for fd, from_asm, lno, ltxt in fifo:
                        if not from_asm:
                            say(f"{name}: line {no}: counted inline-asm wait `{t}` with the compiler-generated request of "
                                f"line {lno} (`{ltxt}`) possibly in flight: the literal no longer counts the ring's requests")
                if n < len(fifo):
                    del fifo[: len(fifo) - n]
        return tuple(fifo)

    seen = [set() for _ in blocks]
    work = [(0, ())]
    nstates = 0
    while work:
        b, fifo = work.pop()
        key = tuple((d, a) for d, a, _, _ in fifo)        # lines are for messages only: two FIFOs that differ in them behave alike
        if key in seen[b]:
            continue
        seen[b].add(key)
        nstates += 1
        if nstates > max_states:
            say(f"{name}: more than {max_states} (block, FIFO) states: not analysed completely")
            break
        out = run_block(b, fifo)
        for s2 in successors(b):
            work.append((s2, out))
    return msgs


# kernels that must not spill or use scratch: everything in the three files (a spill anywhere in them is a regression, and in
# the ring kernels it is a correctness bug)
def check_file(path, text=None):
    asm = text if text is not None else open(path).read()
    msgs = []
    meta = kernel_metadata(asm)
    for k, d in sorted(meta.items()):
        for key in ("private_segment_fixed_size", "vgpr_spill_count"):       # (SGPR spills go to VGPR lanes, not to memory)
            if d.get(key, 0) != 0:
                msgs.append(f"{k}: {key} = {d[key]} (VGPRs {d.get('vgpr_count')}): scratch traffic shifts every counted vmcnt")
    kernels = split_kernels(asm)
    checked = 0
    for name, insts in kernels.items():
        r = check_kernel(name, insts)
        if any(a and is_vmem(t.split()[0]) for _, t, a in insts):
            checked += 1
        msgs.extend(r)
    return msgs, len(meta), checked


def main(argv):
    bad = 0
    for p in argv:
        msgs, nk, nc = check_file(p)
        for m in msgs:
            print("VIOLATION", p, m)
        print(f"{p}: {nk} kernels (metadata), {nc} with inline-asm vector memory simulated, {len(msgs)} violation(s)")
        bad += len(msgs)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
