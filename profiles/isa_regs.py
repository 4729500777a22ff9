"""Register / scratch use of every kernel in an ISA listing: python profiles/isa_regs.py <file.s> [name filter]"""
import re, subprocess, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for m in re.finditer(r"\.amdhsa_kernel (\S+).*?\.end_amdhsa_kernel", s, re.S):
    blk = m.group(0)
    rows.append((m.group(1), re.search(r"next_free_vgpr (\d+)", blk).group(1), re.search(r"next_free_sgpr (\d+)", blk).group(1),
                 re.search(r"private_segment_fixed_size (\d+)", blk).group(1)))
names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.split("\n")
for (n, v, sg, sc), d in zip(rows, names):
    d = d.replace("void jinc::(anonymous namespace)::", "").split("(")[0]
    if flt in d:
        print(f"{d:60s} vgpr {v:>4s} sgpr {sg:>4s} scratch {sc}")
