#!/usr/bin/env python3
"""Registers, spills and scratch of every step-kernel instance IN THE SHIPPED OBJECTS (mpc_quad_ros_amd/csrc/build/*.o or libmpcq.so):
llvm-readelf --notes on the gfx950 code objects.   usage: kernel_resources_of_objects.py file.o|file.so [...]"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_exec_prologue as cep

RD = "/opt/rocm/lib/llvm/bin/llvm-readelf"
print(f"{'object':18s} {'kernel':62s} {'code B':>8s} {'vgpr':>5s} {'agpr':>5s} {'sgpr spill':>10s} {'vgpr spill':>10s} {'scratch B':>9s}")
for path in sys.argv[1:]:
    for _kind, co in cep.code_objects(path):
        sizes = {}
        for line in subprocess.check_output([RD, "-s", "--wide", co]).decode().splitlines():
            f = line.split()
            if len(f) >= 8 and f[3] == "FUNC":
                sizes[f[7]] = int(f[2])
        notes = subprocess.check_output([RD, "--notes", co]).decode()
        for blk in notes.split("  - .agpr_count:")[1:]:
            g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
            name = g("name")
            if "step_kernel" not in name:
                continue
            dem = subprocess.check_output(["c++filt", name]).decode().strip()
            dem = re.sub(r"^void mpcq::", "", dem); dem = re.sub(r"\(.*", "", dem); dem = re.sub(r", (double|float), (true|false)>$", ">", dem)
            agpr = int(blk.split()[0]); total = int(g("vgpr_count"))
            print(f"{os.path.basename(path):18s} {dem[:62]:62s} {sizes.get(name, 0):8d} {total - agpr:5d} {agpr:5d} {g('sgpr_spill_count'):>10s} {g('vgpr_spill_count'):>10s} {g('private_segment_fixed_size'):>9s}")
