#!/usr/bin/env python3
"""Static instruction profile of one column-kernel instantiation by KERNEL source line: every instruction of a
-gline-tables-only build is attributed, through its inlining chain (llvm-symbolizer -i), to the line of
thompson_column_step that it was expanded from.
  hipcc <flags of csrc/Makefile> -gline-tables-only --offload-device-only -c thompson_column.hip -o col.o
  clang-offload-bundler --unbundle --type=o --input=col.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=col.elf
  llvm-objdump -d --no-show-raw-insn col.elf > col.s
usage: isa_lines.py col.elf col.s <mangled-name fragment> [bucket] [regex]
       bucket = lines per histogram bin (default 1); regex = only instructions whose disassembly text matches (re.search)"""
import collections
import re
import subprocess
import sys

elf, path, frag = sys.argv[1:4]
bucket = int(sys.argv[4]) if len(sys.argv) > 4 else 1
only = re.compile(sys.argv[5]) if len(sys.argv) > 5 else None
SYMB = "/opt/rocm/lib/llvm/bin/llvm-symbolizer"
addrs, ops = [], []
inside = False
for line in open(path, errors="ignore"):
    if re.match(r"^[0-9a-f]+ <", line):
        inside = frag in line
        continue
    if not inside:
        continue
    m = re.match(r"^\s+([a-z_0-9]+)\s.*//\s*([0-9A-Fa-f]+):", line)
    if m and (only is None or only.search(line.split("//")[0])):
        ops.append(m.group(1)); addrs.append(int(m.group(2), 16))
out = subprocess.run([SYMB, "-i", "-e", elf, "--output-style=GNU", "-f", "-s"], input="\n".join("0x%x" % a for a in addrs),
                     capture_output=True, text=True).stdout
# GNU style with -i: per address, pairs of lines (function, file:line), innermost first; addresses separated how? use -a
out = subprocess.run([SYMB, "-a", "-i", "-e", elf, "--output-style=GNU", "-f", "-s"], input="\n".join("0x%x" % a for a in addrs),
                     capture_output=True, text=True).stdout
frames = {}
cur = None
lines = out.splitlines()
i = 0
while i < len(lines):
    l = lines[i]
    if l.startswith("0x"):
        cur = int(l, 16); frames[cur] = []; i += 1
        continue
    fn, loc = l, lines[i + 1]
    frames[cur].append((fn, loc)); i += 2
valu = collections.Counter(); allc = collections.Counter(); helper = collections.Counter()
for a, op in zip(addrs, ops):
    fr = frames.get(a, [])
    kl = None
    for fn, loc in fr:                     # outermost frame = the kernel
        pass
    if fr:
        fn, loc = fr[-1]
        m = re.match(r"(.*):(\d+)", loc)
        kl = int(m.group(2)) if m and "thompson_column.hip" in m.group(1) else None
    key = (kl // bucket) * bucket if kl is not None else -1
    allc[key] += 1
    if op.startswith("v_"):
        valu[key] += 1
        if len(fr) > 1:
            helper[re.sub(r"<.*", "", fr[-2][0])] += 1
tot = sum(valu.values())
print("instructions", len(addrs), "VALU", tot)
for k in sorted(valu):
    print("%5d  %5d %5.1f%%  (all %d)" % (k, valu[k], 100.0 * valu[k] / tot, allc[k]))
print("-- VALU by first-level helper --")
for h, n in helper.most_common(40):
    print("%6d %5.1f%%  %s" % (n, 100.0 * n / tot, h))
