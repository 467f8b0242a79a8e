"""Registers, scratch and spill counts of every fan-kernel instance, from hipcc's -save-temps assembly.
usage: python scripts/kernel_resources.py <dir-with-.s>"""
import re, sys
s = open(sys.argv[1] + "/pgr_hip-hip-amdgcn-amd-amdhsa-gfx950.s").read()
md = s[s.index("amdhsa.kernels:"):]
for blk in md.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if "fan_kernel" not in name:
        continue
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)
    m = re.search(r"ILb(\d)ELi(\d)ELi(\d)ELb(\d)E", name)
    print("LT %s ZM %s SAVE %s PERSIST %s" % m.groups(), "vgpr", g("vgpr_count"), "sgpr", g("sgpr_count"), "scratch", g("private_segment_fixed_size"),
          "sgpr spills", g("sgpr_spill_count"), "vgpr spills", g("vgpr_spill_count"))
