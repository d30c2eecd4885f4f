#!/bin/bash
# A/B of library builds on ONE box (development tool): tools/ab.sh "<command>" name1 name2 ... runs the command once per
# tracy_amd/lib_ab/<name>.so copied over tracy_amd/lib/libtracy_hip.so (the last one stays in place).
cmd="$1"; shift
for n in "$@"; do
  cp tracy_amd/lib_ab/$n.so tracy_amd/lib/libtracy_hip.so
  echo "== $n"
  bash -c "$cmd"
done
