"""Static check of the shipped device code for the store-data hazard that LLVM's hazard recogniser does not cover on gfx950.

A VMEM store of more than 64 bits reads its data registers a few cycles after it issues.  gfx9's rule -- one wait state
before a VALU instruction may overwrite them -- has an exception in the ISA manual, and in LLVM, for MUBUF stores whose
soffset operand is an SGPR.  On MI355X that exception does not hold: with `buffer_store_dwordx4 v[6:9], ..., s25 offen`
followed directly by `v_mov_b32 v6, ...` lanes 12-15 of every 16 stored the new value of v6 (round 4: the persistent
Cholesky's granule forwarder, potrf_persist.h PP_STORE16; low words of forwarded values wrong under valid tags whenever no
other wavefront's instruction fell between the two).  This tool disassembles every gfx950 code object in libapgp.so and
reports each wide store (x3 / x4, buffer / global / flat / scratch) whose data registers are written by one of the next
two instructions without an intervening s_nop.

Usage: python tools/check_store_hazard.py [path/to/libapgp.so]      (exit code 1 if anything is found)"""
import os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def code_objects(lib, workdir):
    """the gfx950 code objects bundled in the library's .hip_fatbin section (one bundle per translation unit)"""
    fat = os.path.join(workdir, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = []
    for i, a in enumerate(starts):
        b = starts[i + 1] if i + 1 < len(starts) else len(blob)
        bundle = os.path.join(workdir, "bundle%d.bin" % i)
        open(bundle, "wb").write(blob[a:b])
        co = os.path.join(workdir, "dev%d.co" % i)
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + bundle,
                            "--targets=" + TARGET, "--output=" + co], capture_output=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            out.append(co)
    return out


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def scan(co):
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
    ins, func = [], "?"
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            func = m.group(1)
            continue
        t = line.split("//")[0].strip()
        if t and re.match(r"^[a-z_0-9]+(\s|$)", t):
            ins.append((func, t))
    found, wide = [], 0
    for k, (fn, t) in enumerate(ins):
        m = re.match(r"(buffer|global|flat|scratch)_store_dwordx[34]\s+(.*)", t)
        if not m:
            continue
        wide += 1
        ops = [o.strip() for o in m.group(2).split(",")]
        data = regs(ops[0]) if m.group(1) == "buffer" else regs(ops[1]) if len(ops) > 1 else set()
        if not data:
            continue
        nops = 0
        for d in (1, 2):
            if k + d >= len(ins) or ins[k + d][0] != fn:
                break
            nt = ins[k + d][1]
            if nt.startswith("s_nop"):
                nops += 1 + int(nt.split()[1])
                continue
            if nops >= 1:
                break
            mm = re.match(r"(v_[a-z0-9_]+|ds_read[a-z0-9_]*|buffer_load[a-z0-9_]*|global_load[a-z0-9_]*|scratch_load[a-z0-9_]*|flat_load[a-z0-9_]*)\s+([^,\s]+)", nt)
            if mm and not mm.group(1).startswith(("v_cmp", "v_readlane", "v_readfirstlane")) and regs(mm.group(2)) & data:
                # (a compiler-generated spill -- scratch_store -- whose registers are the destination of a MEMORY read: the
                # read's data comes back tens of cycles after it issues (LDS >= 64), the store has read its own long before;
                # the hazard seen on the chip was a VALU write in the next issue slot.  Round 5: the D > 16 sweep spills.)
                if m.group(1) == "scratch" and not mm.group(1).startswith("v_"):
                    break
                found.append("%s: `%s` then (+%d) `%s`" % (fn, t, d, nt))
                break
    return wide, found


def check(lib):
    with tempfile.TemporaryDirectory() as wd:
        cos = code_objects(lib, wd)
        if not cos:
            raise RuntimeError("no gfx950 code object found in " + lib)
        wide, found = 0, []
        for co in cos:
            w, f = scan(co)
            wide += w
            found += f
    return len(cos), wide, found


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "approxposterior_amd", "csrc", "libapgp.so")
    n, wide, found = check(lib)
    print("%s: %d code objects, %d wide stores, %d with their data registers overwritten within two issue slots" % (lib, n, wide, len(found)))
    for f in found[:40]:
        print("  " + f)
    sys.exit(1 if found else 0)
