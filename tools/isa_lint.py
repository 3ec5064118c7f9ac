"""Static checks on the gfx950 ISA of libtfft.so (CPU only: llvm-objdump on the embedded code object).

Why: round 1 met an intermittent wrong twiddle product in the column kernel that went away when clang's SLP
vectoriser was switched off. This tool makes the properties the fix relies on checkable without a GPU:

  1. no packed fp32 arithmetic (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) in any kernel that issues MFMAs, apart from
     an explicit allow-list (prologue code that runs before the first MFMA);
  2. the MFMA -> consumer wait states of the gfx950 tables are present in straight-line code: an instruction that reads
     or overwrites the destination VGPRs of a v_mfma_f32_16x16x32_f16 (4 passes) must be at least 8 wait states behind
     it when it is a VALU / memory / LDS instruction or another MFMA reading them as A/B, 6 when it is an MFMA reading
     them as C. (Instructions count one wait state each, `s_nop N` counts N + 1. LLVM's GCNHazardRecognizer inserts these
     for code it schedules; inline asm is not covered by it, and this check is independent of it.)

  3. no wave ends with an LDS-DMA (global_load_lds_*) possibly in flight: on every control-flow path from such a load to an
     s_endpgm lies an `s_waitcnt vmcnt(0)` (checked on the disassembly's control-flow graph; pins the fix of commit 6ed7571).

usage: python tools/isa_lint.py [libtfft.so | file.s | disassembly.txt]   (exit code 1 on a finding)
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM_BIN = "/opt/rocm/lib/llvm/bin"

# kernel-name substring -> number of packed fp32 instructions tolerated, with the reason
PK_ALLOW = {}

MFMA_PASSES = {"v_mfma_f32_16x16x32_f16": 4, "v_mfma_f32_16x16x32_bf16": 4, "v_mfma_f32_32x32x16_f16": 8,
               "v_mfma_f32_32x32x16_bf16": 8, "v_mfma_f32_16x16x16_f16": 4}


def disassemble(so_path):
    """Text disassembly of the gfx950 code object embedded in a HIP shared library."""
    tmp = tempfile.mkdtemp(prefix="tfft_isa_")
    try:
        local = os.path.join(tmp, os.path.basename(so_path))
        shutil.copy(so_path, local)
        subprocess.check_call([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", local], cwd=tmp,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cos = [f for f in os.listdir(tmp) if "amdgcn" in f and "gfx950" in f]
        if not cos:
            raise RuntimeError("no gfx950 code object found in " + so_path)
        return subprocess.check_output([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", os.path.join(tmp, cos[0])], text=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


_REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def vregs(operand):
    out = set()
    for m in _REG.finditer(operand):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def split_kernels(text):
    """{kernel symbol: [instruction lines]} from a .s file or an llvm-objdump listing."""
    kernels, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^(?:[0-9a-f]+ <)?(_Z\w+)>?:", line)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        if cur is None:
            continue
        s = line.strip()
        if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
            if s.endswith(":") and not s.startswith("."):
                kernels[cur].append("@label")
            elif re.match(r"^\.?L?BB\d+_\d+:", s):
                kernels[cur].append("@label")
            continue
        s = s.split("//")[0].split(";")[0].strip()
        if s:
            kernels[cur].append(s)
    return kernels


def lint_kernel(name, insts):
    findings = []
    n_mfma = sum(1 for i in insts if i.startswith("v_mfma"))
    pk = [i for i in insts if re.match(r"v_pk_(mul|fma|add)_f32", i)]
    if n_mfma and pk:
        allowed = max([v[0] for k, v in PK_ALLOW.items() if k in name] or [0])
        if len(pk) > allowed:
            findings.append(f"{len(pk)} packed fp32 instruction(s) in an MFMA kernel (allowed {allowed}), first: {pk[0]}")
    # wait-state check
    inflight = []   # [dst regs, wait states elapsed, passes, text]
    for ins in insts:
        if ins == "@label":
            continue
        op, _, rest = ins.partition(" ")
        ws = 1
        if op == "s_nop":
            try:
                ws = int(rest.strip(), 0) + 1
            except ValueError:
                ws = 1
        is_vector = op.startswith(("v_", "ds_", "global_", "buffer_", "flat_", "scratch_"))
        if is_vector and inflight:
            ops = [o.strip() for o in rest.split(",")]
            touched = vregs(rest)
            for dst, elapsed, passes, text in inflight:
                if not (touched & dst):
                    continue
                need = passes + 4            # gfx950: XDL write VGPR -> VALU / VMEM / LDS read or write, MFMA SrcA/B read
                if op.startswith("v_mfma") and len(ops) >= 4:
                    ab = vregs(ops[1]) | vregs(ops[2])
                    c_or_d = vregs(ops[3]) | vregs(ops[0])
                    if not (ab & dst) and (c_or_d & dst):
                        # same-register accumulate chain needs none; overlapped-but-different SrcC needs passes + 2
                        need = 0 if (vregs(ops[3]) == dst and vregs(ops[0]) == dst) else passes + 2
                if elapsed < need:
                    findings.append(f"{need} wait states needed, {elapsed} present: `{text}` -> `{ins}`")
        for e in inflight:
            e[1] += ws
        inflight = [e for e in inflight if e[1] < 24]
        if op in MFMA_PASSES:
            inflight.append([vregs(rest.split(",")[0]), 0, MFMA_PASSES[op], ins])
    return n_mfma, len(pk), findings


_ADDR = re.compile(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]{8,}):")


def split_kernels_addr(text):
    """{kernel symbol: [(address, instruction text)]} from an llvm-objdump listing (needs the `// ADDRESS: ENCODING`
    comments; a plain .s file yields nothing)."""
    kernels, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(_Z\w+)>:", line)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        if cur is None:
            continue
        m = _ADDR.match(line)
        if m:
            kernels[cur].append((int(m.group(2), 16), m.group(1).strip()))
    return kernels


def _sgprs(operand):
    out = set()
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b", operand):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def lint_dma_drain(insts):
    """3. A wave must not end with LDS-DMA in flight (global_load_lds_* writes LDS asynchronously; a workgroup that retires
    gives its LDS to the next one while the transfer may still land). Exploration of the kernel's control-flow graph
    with the abstract state ("an LDS-DMA may be outstanding", known loop-exit flags): global_load_lds sets the first,
    `s_waitcnt vmcnt(0)` clears it (a partial wait does not); every s_endpgm must be reached with it clear on ALL paths.
    The one piece of path sensitivity: the structuriser's loop-latch idiom
        exit path:  s_mov_b64 s[a:b], -1      continue path:  s_mov_b64 s[a:b], 0
        latch:      s_andn2_b64 vcc, exec, s[a:b] ; s_cbranch_vccz <exit>
    is followed exactly (an SGPR pair set to 0 / -1 by s_mov_b64 is tracked until something else writes it; exec is taken
    to be non-zero), because "issue the look-ahead copy, then loop" and "leave" are merged into one latch block by the
    compiler and only that flag tells them apart. Returns findings."""
    if not any(t.startswith("global_load_lds") for _, t in insts):
        return []
    index = {a: i for i, (a, _) in enumerate(insts)}
    n = len(insts)

    def branch_target(a, rest):
        try:
            off = int(rest.strip().split()[0], 0) & 0xffff
        except (ValueError, IndexError):
            return None
        if off >= 0x8000:
            off -= 0x10000
        return index.get(a + 4 + 4 * off)

    bad = set()
    seen = set()
    work = [(0, False, frozenset())]
    while work:
        st = work.pop()
        if st in seen:
            continue
        seen.add(st)
        i, dma, facts = st
        a, t = insts[i]
        op, _, rest = t.partition(" ")
        if op == "s_endpgm":
            if dma:
                bad.add(a)
            continue
        f = dict(facts)
        if t.startswith("global_load_lds"):
            dma = True
        elif op == "s_waitcnt" and re.search(r"vmcnt\(0\)", t):
            dma = False
        ops = [o.strip() for o in rest.split(",")]
        taken = fall = True
        if op in ("s_cbranch_vccz", "s_cbranch_vccnz") and "vcc" in f:
            zero = f["vcc"] == "z"
            taken = zero if op == "s_cbranch_vccz" else not zero
            fall = not taken
        elif op == "s_mov_b64" and len(ops) == 2 and re.fullmatch(r"s\[\d+:\d+\]", ops[0]) and ops[1] in ("0", "-1"):
            f[ops[0]] = ops[1]
        elif op in ("s_andn2_b64", "s_and_b64") and len(ops) == 3 and ops[0] == "vcc" and ops[1] == "exec" and ops[2] in f:
            nz = (f[ops[2]] == "0") if op == "s_andn2_b64" else (f[ops[2]] == "-1")
            f["vcc"] = "nz" if nz else "z"
        elif op.startswith(("s_", "v_cmp", "v_readfirstlane", "v_readlane")) and not op.startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_barrier")):
            # anything else that may write a tracked SGPR pair or vcc: forget it (destination = first operand; v_cmp writes vcc)
            dst = ops[0] if ops else ""
            wr = _sgprs(dst)
            for k in list(f):
                if k == "vcc":
                    if "vcc" in dst or op.startswith("v_cmp"):
                        del f[k]
                elif _sgprs(k) & wr:
                    del f[k]
        nf = frozenset(f.items())
        if op == "s_branch":
            tgt = branch_target(a, rest)
            if tgt is not None:
                work.append((tgt, dma, nf))
            continue
        if op.startswith("s_cbranch"):
            tgt = branch_target(a, rest)
            if tgt is not None and taken:
                work.append((tgt, dma, nf))
            if fall and i + 1 < n:
                work.append((i + 1, dma, nf))
            continue
        if i + 1 < n:
            work.append((i + 1, dma, nf))
    return [f"s_endpgm at {a:#x} reachable with an LDS-DMA (global_load_lds) possibly in flight: no s_waitcnt vmcnt(0) on some path"
            for a in sorted(bad)]


def lint_text(text):
    report = {}
    with_addr = split_kernels_addr(text)
    for name, insts in split_kernels(text).items():
        n_mfma, n_pk, findings = lint_kernel(name, insts)
        n_dma = sum(1 for i in insts if i.startswith("global_load_lds"))
        findings = findings + lint_dma_drain(with_addr.get(name, []))
        report[name] = {"mfma": n_mfma, "pk_f32": n_pk, "lds_dma": n_dma, "findings": findings}
    return report


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tensor-fft_amd", "libtfft.so")
    text = disassemble(path) if path.endswith(".so") else open(path).read()
    rep = lint_text(text)
    bad = 0
    for name, r in sorted(rep.items()):
        if r["mfma"] or r["findings"]:
            print(f"{name[:110]:110s} mfma {r['mfma']:4d} pk_f32 {r['pk_f32']:4d} findings {len(r['findings'])}")
        for f in r["findings"][:8]:
            print("    " + f)
        bad += len(r["findings"])
    print(f"{len(rep)} kernels, {bad} finding(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
