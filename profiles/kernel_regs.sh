#!/bin/sh
# Registers, spills, LDS and occupancy of every kernel of one source file of the library, as the compiler reports them:
#   profiles/kernel_regs.sh rdr_geometry.hip [extra compiler flags]
cd "$(dirname "$0")/../renderer_amd/csrc" || exit 1
src="$1"; shift
# the flags the Makefile gives the file on top of the common ones
case "$src" in
  rdr_shade.hip) set -- -fno-slp-vectorize -fno-hip-fp32-correctly-rounded-divide-sqrt -DRDR_FAST_DIVISION "$@" ;;
  rdr_shadow.hip) set -- -fno-slp-vectorize "$@" ;;
esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math --cuda-device-only \
    -Rpass-analysis=kernel-resource-usage "$@" -c "$src" -o /dev/null 2>&1 |
  awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name)}
       /VGPRs:/ {v=$0; sub(/.*VGPRs: /,"",v); sub(/ \[.*/,"",v)}
       /ScratchSize/ {s=$0; sub(/.*: /,"",s); sub(/ \[.*/,"",s)}
       /Occupancy/ {o=$0; sub(/.*: /,"",o); sub(/ \[.*/,"",o)}
       /LDS Size/ {l=$0; sub(/.*: /,"",l); sub(/ \[.*/,"",l); printf "%-90s vgpr %4s scratch %6s occupancy %2s lds %6s\n", name, v, s, o, l}'
