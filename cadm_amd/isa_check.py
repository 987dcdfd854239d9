"""Post-build scan of the generated gfx950 code of the hand-scheduled kernels (no GPU needed, llvm-objdump only).

Why the product carries this and not only the test-suite: the rollout kernels name register classes and pad hazards by hand --
register-resident weight fragments are pinned to AGPRs through inline-asm constraints, every MFMA of a sweep is inline asm, and the
wait states hipcc's hazard recognizer cannot see through an asm statement are written out as `s_nop`s (csrc/rollout_xdl.h).  Whether
those assumptions hold is a property of the COMPILER THAT BUILT THE CODE: under register pressure hipcc parks values in spare AGPRs and
copies them back right in front of the statement that reads them.  The library is built (and scanned by tests/test_isa_hygiene.py) with
one hipcc; a user's geometry is built on demand (cadm_amd/jit.py) with whatever hipcc that machine has.  So `jit.build` runs the same
rules on every module it compiles and refuses to register one that breaks them.

Rules (each returns human-readable problems; an empty list = clean):
  rollout_xdl_kernel (asm-MFMA geometries: those that load resident fragments)
    R1  a resident-fragment AGPR is never copied (v_accvgpr_read / v_accvgpr_write) and no MFMA reads a non-resident AGPR operand
    R2  an SGPR offset written by v_readfirstlane / v_readlane has >= 5 wait states before the asm buffer load that reads it
    R3  every asm MFMA has its own `s_nop >= 1` in front (VALU-written VGPR -> MFMA read)
    (scratch traffic is reported separately: a performance defect, compiler-managed and hazard-safe)
  rollout_wt_kernel
    W1  the weights move by LDS-DMA (`buffer_load_dwordx4 .. lds`)
    W2  walking back from every s_barrier, an `s_waitcnt vmcnt(0)` comes before any LDS-DMA request
  chain_kernel (training; library only)
    T1  from the first ring load on -- following CONTROL FLOW, not the image's block layout -- nothing but the asm ring loads and the asm
        MFMAs touches a0-a63
    T2  no VALU instruction directly in front of an asm MFMA; MFMA / ring-load operand shapes are the ones the source names
"""
import os
import re
import shutil
import struct
import subprocess
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


class IsaCheckError(RuntimeError):
    pass


def find_objdump(hipcc=None):
    """llvm-objdump of the ROCm installation the compiler belongs to ($CADM_OBJDUMP overrides), else whatever is on PATH; None if absent."""
    cands = [os.environ.get("CADM_OBJDUMP")]
    if hipcc:
        root = os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
        cands += [os.path.join(root, "lib", "llvm", "bin", "llvm-objdump"), os.path.join(root, "llvm", "bin", "llvm-objdump"),
                  os.path.join(os.path.dirname(os.path.realpath(hipcc)), "llvm-objdump")]
    cands += [os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin", "llvm-objdump"),
              "/opt/rocm/lib/llvm/bin/llvm-objdump", shutil.which("llvm-objdump")]
    return next((c for c in cands if c and os.path.exists(c)), None)


def code_objects(path):
    """gfx950 ELF images inside a shared object's / object file's .hip_fatbin (clang offload bundles, one per translation unit)."""
    data = open(path, "rb").read()
    out, pos = [], 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            return out
        n = struct.unpack_from("<Q", data, i + len(MAGIC))[0]
        off = i + len(MAGIC) + 8
        for _ in range(n):
            o, s, ts = struct.unpack_from("<QQQ", data, off)
            triple = data[off + 24:off + 24 + ts]
            off += 24 + ts
            if b"gfx950" in triple and s > 0:
                out.append(data[i + o:i + o + s])
        pos = i + len(MAGIC)


def kernels(image, match, objdump):
    """{symbol: [instruction line, ..]} of the functions of one code object whose (mangled) name contains `match`.  A line is llvm-objdump's:
    `mnemonic operands   // ADDRESS: ENCODING [<symbol+0xOFFSET>]` (the trailing comment carries the address and a branch's target)."""
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(image)
        f.flush()
        txt = subprocess.run([objdump, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
    out, name, buf = {}, None, []
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if name is not None:
                out[name] = buf
            name, buf = (m.group(1), []) if match in m.group(1) else (None, [])
        elif name is not None and line.strip():
            buf.append(line.strip())
    if name is not None:
        out[name] = buf
    return out


def strip(ins):
    """instruction text without the trailing address comment"""
    return [x.split("//")[0].strip() for x in ins if x.strip()]


def _addr(line):
    m = re.search(r"//\s*([0-9A-Fa-f]+):", line)
    return int(m.group(1), 16) if m else None


def _successors(ins):
    """per-instruction successor lists (fall-through + direct branches), or None when the function has an indirect jump / a branch whose
    target cannot be decoded: the caller then takes the conservative answer"""
    addr = [_addr(x) for x in ins]
    base = addr[0]
    at = {a: i for i, a in enumerate(addr) if a is not None}
    text = strip(ins)
    if any(t.startswith(("s_setpc", "s_swappc")) for t in text):
        return None
    succ = []
    for i, t in enumerate(text):
        nxt = []
        if t.startswith(("s_branch", "s_cbranch")):
            m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", ins[i])
            tgt = at.get(base + int(m.group(1), 16)) if m else None
            if tgt is None:
                return None
            nxt.append(tgt)
        if not t.startswith(("s_branch", "s_endpgm")) and i + 1 < len(ins):
            nxt.append(i + 1)
        succ.append(nxt)
    return succ


def _reach(succ, start, avoid=-1):
    seen, stack = set(), [start]
    while stack:
        i = stack.pop()
        if i in seen or i == avoid:
            continue
        seen.add(i)
        stack.extend(succ[i])
    return seen


def reachable_from(ins, start):
    """Indices of the instructions reachable from instruction `start` by control flow (fall-through + direct branches).  An indirect
    jump (s_setpc / s_swappc) makes everything reachable: the conservative answer."""
    succ = _successors(ins)
    if succ is None:
        return set(range(len(ins)))
    return _reach(succ, start)


def dominated_by(ins, node):
    """Indices of the instructions DOMINATED by instruction `node`: every path from the function's entry to them passes through it
    (reachable from the entry, but not once `node` is taken out of the graph).  This is "the code that runs with what `node` did in
    effect" without knowing the values a branch tests: hipcc merges the tails of the kernel's wave variants into shared blocks guarded by
    flags, so plain reachability leaks from one variant into the next; dominance does not."""
    succ = _successors(ins)
    if succ is None:
        return set(range(len(ins)))
    return (_reach(succ, 0) - _reach(succ, 0, avoid=node)) | {node}


# ---------------------------------------------------------------------------------------------------------------------------------
# rollout_xdl_kernel
# ---------------------------------------------------------------------------------------------------------------------------------
def xdl_resident_loads(ins):
    """indices of the resident-fragment loads: the asm statement is "s_nop 4; buffer_load_dwordx4 a[..]" -- that pair identifies them
    (hipcc may also point ordinary ring loads at spare AGPRs, which it tracks itself)"""
    return [i for i, x in enumerate(ins) if i > 0 and x.startswith("buffer_load_dwordx4 a[") and ins[i - 1].startswith("s_nop 4")]


def check_rollout_xdl(sym, raw):
    """(problems, info) for one rollout_xdl_kernel.  Geometries without resident fragments use builtin MFMAs (hipcc handles every hazard):
    only their scratch use is reported."""
    ins = strip(raw)
    problems = []
    info = {"mfma": sum(1 for x in ins if x.startswith("v_mfma_f32_16x16x32_f16")), "scratch": sum(1 for x in ins if x.startswith("scratch_"))}
    loads = xdl_resident_loads(ins)
    info["resident_loads"] = len(loads)
    if not loads:
        return problems, info
    mfma_at = [i for i, x in enumerate(ins) if x.startswith("v_mfma_")]
    # The kernel body exists once per wave variant (tiles per wave: a scalar branch at the kernel's entry picks one, they meet again at
    # s_endpgm); a variant with resident fragments loads them with a burst of loads in its prologue and keeps them to the end.  What a burst
    # pins is decided by CONTROL FLOW: the instructions DOMINATED by the burst's first load (every path from the kernel's entry to them
    # runs through it) are that variant's "fragments are live" region -- a variant WITHOUT resident fragments (one hidden layer: nothing to
    # pin on the waves that own no head tile) is not dominated by another variant's burst and may use any AGPR as spill space, as may the
    # prologue in front of a burst.  (Plain reachability is too coarse: hipcc merges the variants' tails into shared, flag-guarded blocks.)
    bursts = []
    for k, i in enumerate(loads):
        if k == 0 or any(loads[k - 1] < m < i for m in mfma_at):
            bursts.append(i)
    covered = set()
    for b0 in bursts:
        live = dominated_by(raw, b0)
        covered |= live
        resident = set()
        for i in loads:
            if i in live:
                m2 = re.match(r"buffer_load_dwordx4 a\[(\d+):(\d+)\]", ins[i])
                resident.update(range(int(m2.group(1)), int(m2.group(2)) + 1))
        for i in sorted(live):
            x = ins[i]
            m2 = re.match(r"v_accvgpr_write_b32 a(\d+),", x) or re.match(r"v_accvgpr_read_b32 v\d+, a(\d+)", x)
            if m2 and int(m2.group(1)) in resident:
                problems.append("%s: R1 resident fragment register copied: %s" % (sym, x))
            m2 = re.match(r"v_mfma_f32_16x16x32_f16 v\[\d+:\d+\], a\[(\d+):(\d+)\], v\[", x)
            if m2 and not set(range(int(m2.group(1)), int(m2.group(2)) + 1)) <= resident:
                problems.append("%s: R1 MFMA reads an AGPR operand that is not a resident fragment: %s" % (sym, x))
    # an asm MFMA that reads an AGPR operand outside every burst's region has no fragment to read
    for i, x in enumerate(ins):
        if i not in covered and re.match(r"v_mfma_f32_16x16x32_f16 v\[\d+:\d+\], a\[", x):
            problems.append("%s: R1 MFMA reads an AGPR operand on a path without resident loads: %s" % (sym, x))
    for i in loads:
        m2 = re.match(r"buffer_load_dwordx4 a\[\d+:\d+\], v\d+, s\[\d+:\d+\], (s\d+) offen", ins[i])
        if not m2:
            continue
        waits = 0
        for y in reversed(ins[max(0, i - 6):i]):
            if re.match(r"v_read(first)?lane_b32 %s," % m2.group(1), y):
                if waits < 5:
                    problems.append("%s: R2 %s only %d wait states after %s" % (sym, ins[i], waits, y))
                break
            m3 = re.match(r"s_nop (\d+)", y)
            waits += int(m3.group(1)) + 1 if m3 else 1
    for i, x in enumerate(ins):
        if x.startswith("v_mfma_f32_16x16x32_f16"):
            if not (ins[i - 1].startswith("s_nop") and int(ins[i - 1].split()[1]) >= 1):
                problems.append("%s: R3 MFMA without wait states in front: %s / %s" % (sym, ins[i - 1], x))
    info["resident_mfma"] = sum(1 for x in ins if x.startswith("v_mfma_f32_16x16x32_f16") and re.search(r", a\[\d+:\d+\], v\[", x))
    if info["resident_mfma"] <= 10:
        problems.append("%s: resident fragments are loaded but no MFMA reads its A operand from AGPRs" % sym)
    return problems, info


# ---------------------------------------------------------------------------------------------------------------------------------
# rollout_wt_kernel
# ---------------------------------------------------------------------------------------------------------------------------------
def check_rollout_wt(sym, raw):
    ins = strip(raw)
    problems = []
    info = {"mfma": sum(1 for x in ins if x.startswith("v_mfma_f32_16x16x32_f16")), "scratch": sum(1 for x in ins if x.startswith("scratch_"))}
    if not any(x.startswith("buffer_load_dwordx4") and x.endswith(" lds") for x in ins):
        problems.append("%s: W1 no LDS-DMA weight requests" % sym)
    # the weight ring is only correct if every wave's LDS-DMA pieces (tracked by vmcnt) have landed BEFORE the block barrier
    for i, x in enumerate(ins):
        if not x.startswith("s_barrier"):
            continue
        j = i - 1
        ok = True
        while j >= 0 and not (ins[j].startswith("s_waitcnt") and re.search(r"vmcnt\(0\)", ins[j])):
            if ins[j].startswith("buffer_load_dwordx4") and ins[j].endswith(" lds"):
                problems.append("%s: W2 an LDS-DMA request at instruction %d reaches the barrier at %d without a vmcnt(0) wait" % (sym, j, i))
                ok = False
                break
            if ins[j].startswith(("s_barrier", "s_endpgm")):
                problems.append("%s: W2 barrier at %d is not preceded by s_waitcnt vmcnt(0)" % (sym, i))
                ok = False
                break
            j -= 1
        if ok and j < 0:
            problems.append("%s: W2 barrier at %d is not preceded by s_waitcnt vmcnt(0)" % (sym, i))
    return problems, info


# ---------------------------------------------------------------------------------------------------------------------------------
# chain_kernel (training)
# ---------------------------------------------------------------------------------------------------------------------------------
def check_chain(sym, raw):
    """The training chains' operand ring is a[0:63], named literally in inline asm (csrc/train.hip).  From the first ring load on, a0-a63
    hold operand blocks between one asm statement (the load) and another (the MFMAs): a value hipcc parks there would overwrite them.  In
    front of it -- the kernel prologue -- the ring is empty and hipcc may use the registers as it likes.  "From .. on" is decided by CONTROL
    FLOW: a block placed earlier in the image but executed after the ring prologue (a loop tail) counts."""
    ins = strip(raw)
    problems = []
    info = {"scratch": sum(1 for x in ins if x.startswith("scratch_"))}
    mine = ("v_mfma_f32_16x16x4_f32", "global_load_dwordx4 a[")
    ring_at = [i for i, x in enumerate(ins) if x.startswith("global_load_dwordx4 a[")]
    if not ring_at:
        problems.append("%s: no ring loads" % sym)
        return problems, info
    live = reachable_from(raw, min(ring_at))
    info["reachable"] = len(live)
    info["before_ring_only"] = len(ins) - len(live)
    for i in sorted(live):
        x = ins[i]
        if x.startswith(mine):
            continue
        for m in re.finditer(r"\ba\[?(\d+)(?::(\d+))?", x):
            if int(m.group(1)) < 64:
                problems.append("%s: T1 hipcc touches the ring's AGPRs (a0-a63) on a path behind the first ring load: %s" % (sym, x))
    mf = [i for i, x in enumerate(ins) if x.startswith("v_mfma_")]
    info["mfma"] = len(mf)
    for i in mf:
        if not re.match(r"v_mfma_f32_16x16x4_f32 v\[\d+:\d+\], a\d+, v\d+, v\[", ins[i]):
            problems.append("%s: T2 unexpected MFMA shape: %s" % (sym, ins[i]))
        if not ins[i - 1].split()[0].startswith(("s_", "v_mfma")):
            problems.append("%s: T2 VALU instruction in front of an asm MFMA: %s / %s" % (sym, ins[i - 1], ins[i]))
    for x in (ins[i] for i in ring_at):
        if not re.match(r"global_load_dwordx4 a\[\d+:\d+\], v\d+, s\[\d+:\d+\]", x):
            problems.append("%s: T2 unexpected ring load: %s" % (sym, x))
    return problems, info


CHECKS = (("rollout_xdl_kernel", check_rollout_xdl), ("rollout_wt_kernel", check_rollout_wt), ("chain_kernel", check_chain))


def scan(path, objdump=None, which=("rollout_xdl_kernel", "rollout_wt_kernel", "chain_kernel")):
    """Run every rule on every matching kernel of a built library / JIT module.  Returns
    {"kernels": n, "problems": [..], "scratch": {sym: count}, "objdump": path}."""
    objdump = objdump or find_objdump()
    if not objdump:
        raise IsaCheckError("llvm-objdump not found ($CADM_OBJDUMP, <rocm>/lib/llvm/bin, PATH)")
    rep = {"kernels": 0, "problems": [], "scratch": {}, "objdump": objdump, "per_kernel": {}}
    for img in code_objects(path):
        for match, fn in CHECKS:
            if match not in which:
                continue
            for sym, raw in kernels(img, match, objdump).items():
                problems, info = fn(sym, raw)
                rep["kernels"] += 1
                rep["problems"] += problems
                rep["per_kernel"][sym] = info
                if info.get("scratch"):
                    rep["scratch"][sym] = info["scratch"]
    return rep
