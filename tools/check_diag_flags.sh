#!/bin/bash
# Exit 1 (and say why) if any -D<switch> of the diagnostic families among the arguments is not listed in tools/diag_switches.txt.
here="$(dirname "$0")"
rc=0
for w in "$@"; do
  case "$w" in
    -DMF_*|-DOHGPU_DIAG*|-DOHGPU_WG_*|-DOHGPU_PLAN_*|-DOHGPU_LINE_*|-DOHGPU_LEGACY*)
      name=${w#-D}; name=${name%%=*}
      if ! grep -q "^$name[[:space:]]\+\(unchanged\|checked\)" "$here/diag_switches.txt"; then
        echo "refused: $name is not listed in tools/diag_switches.txt (say what it does to the kernel's addressing there first)" >&2
        rc=1
      fi;;
  esac
done
exit $rc
