#!/bin/bash
# Code size and register use of the step-kernel instances (no GPU needed): device-only compile of one translation unit,
# then llvm-readelf on the gfx950 code object.   usage: tools/kernel_resources.sh [spec|api] [extra hipcc flags...]
set -e
UNIT=${1:-spec}; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/mpc_quad_ros_amd/csrc
TMP=$(mktemp -d)
if [ "$UNIT" = spec ]; then OPT="-O3 -DMPCQ_UNROLL_FACTOR=2 -DMPCQ_UNROLL_SWEEP=10"; F=mpcq_spec.hip; else OPT="-O2"; F=mpcq_api.hip; fi
/opt/rocm/bin/hipcc -fno-strict-aliasing -std=c++17 --offload-arch=gfx950 ${VGPRFORM--mllvm -amdgpu-mfma-vgpr-form=1} $OPT "$@" --cuda-device-only -c -o $TMP/dev.o $SRC/$F
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$TMP/dev.o --output=$TMP/dev.co --unbundle
LLVM=/opt/rocm/lib/llvm/bin
python3 - "$TMP/dev.co" <<'PY'
import re, subprocess, sys
co = sys.argv[1]
rd = "/opt/rocm/lib/llvm/bin/llvm-readelf"
sizes = {}
for line in subprocess.check_output([rd, "-s", "--wide", co]).decode().splitlines():
    f = line.split()
    if len(f) >= 8 and f[3] == "FUNC":
        sizes[f[7]] = int(f[2])
notes = subprocess.check_output([rd, "--notes", co]).decode()
print(f"{'kernel':58s} {'code B':>8s} {'vgpr':>5s} {'agpr':>5s} {'sgpr spill':>10s} {'vgpr spill':>10s} {'scratch B':>9s}")
for blk in notes.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
    name = g("name")
    dem = subprocess.check_output(["c++filt", name]).decode().strip()
    dem = re.sub(r"^void mpcq::", "", dem)
    dem = re.sub(r"\(.*", "", dem)
    dem = re.sub(r", (double|float), (true|false)>$", ">", dem)
    print(f"{dem[:58]:58s} {sizes.get(name, 0):8d} {g('vgpr_count'):>5s} {blk.split()[0]:>5s} {g('sgpr_spill_count'):>10s} {g('vgpr_spill_count'):>10s} {g('private_segment_fixed_size'):>9s}")
PY
rm -rf $TMP
