#!/usr/bin/env python3
"""Static instruction count of a simulator kernel per inlined stage (development aid; no GPU needed).

Compiles hoic_capi.hip to gfx950 assembly with line tables (-g1), takes one kernel's body and attributes every instruction
to the chain of inlined device functions the `.loc ... @[ ... ]` comments name (outermost first, three levels).  Most stages
are straight-line code per pass, so static count x passes is close to the dynamic count (163 k VALU instructions per env-step
by SQ_INSTS_VALU); branches that are not taken (narrow-phase routines of pair types that do not occur) make it an upper bound.

usage: python3 tools/static_insn_count.py [kernel-name-prefix, default _Z19hoic_substep_kernelILi2E] [opcode prefixes, comma list]
With opcode prefixes (e.g. v_mov,v_cndmask) a third column counts only those instructions, and the table is sorted by it."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hoic_amd", "csrc")
kernel = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else "_Z19hoic_substep_kernelILi2E"
ops = tuple(sys.argv[2].split(",")) if len(sys.argv) > 2 else None
asm = os.path.join(tempfile.gettempdir(), "hoic_capi_g1.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-mllvm", "-amdgpu-mfma-vgpr-form",
                       "-Wno-unused-value", "-S", "--cuda-device-only", "-g1", "-o", asm, "hoic_capi.hip"], cwd=CSRC)
lines, on = [], False
for l in open(asm):
    if l.startswith(kernel) and l.rstrip().endswith(":") or (l.startswith(kernel) and ":" in l.split(";")[0]):
        on = True
    if on:
        lines.append(l.rstrip("\n"))
        if "s_endpgm" in l:
            break


def func_starts(path):
    out = []
    for i, l in enumerate(open(path), 1):
        m = re.match(r"^(?:template\s*<[^>]*>\s*)?(?:__device__|HD|__global__|static)\b.*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", l)
        if m and not l.strip().startswith("//"):
            out.append((i, m.group(1)))
    return out


fr = {f: func_starts(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))}


def fn(file, line):
    name = file
    for l, n in fr.get(file, []):
        if l <= line:
            name = n
        else:
            break
    return name


cnt, cnt_v, cnt_o, chain = collections.Counter(), collections.Counter(), collections.Counter(), None
for l in lines:
    if ".loc" in l and ";" in l:
        chain = [(m.group(1), int(m.group(2))) for m in re.finditer(r"(?:\./)?([\w\.]+):(\d+):\d+", l.split(";", 1)[1])]
        continue
    m = re.match(r"^\t([a-z_0-9]+)\s", l)
    if not m or chain is None or m.group(1).startswith("s_nop"):
        continue
    names = [fn(f, ln) for f, ln in chain]
    lvl = [n for n in reversed(names) if n.startswith(("dev_", "col_", "hull_", "path_gather", "hs_", "wave_", "lc_", "make_frame", "load_state", "store_state"))]
    key = " > ".join(lvl[:3]) if lvl else "kernel body"
    cnt[key] += 1
    if m.group(1).startswith("v_"):
        cnt_v[key] += 1
    if ops and m.group(1).startswith(ops):
        cnt_o[key] += 1
print(f"{kernel}: {sum(cnt.values())} instructions, {sum(cnt_v.values())} VALU")
if ops:
    print(f"{sum(cnt_o.values())} of them {'/'.join(ops)}")
    for k, v in cnt_o.most_common(45):
        print(f"{cnt[k]:6d} {cnt_v[k]:6d} {v:6d}  {k}")
else:
    for k, v in cnt.most_common(45):
        print(f"{v:6d} {cnt_v[k]:6d}  {k}")
