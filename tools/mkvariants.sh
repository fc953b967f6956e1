#!/bin/bash
# tools/mkvariants.sh "name:-DFLAG=1 -DOTHER=2" ...  -> tools/_build/libvptq_hip_<name>.so (object directories removed again:
# they would travel to the GPU box with every gpurun call)
for v in "$@"; do
  n=${v%%:*}; e=${v#*:}
  make -C vptq_amd/csrc -j8 variant NAME=$n EXTRA="$e" 2>&1 | grep -E " error|Error " -A3
  rm -rf tools/_build/${n}_obj
done
ls tools/_build/*.so 2>/dev/null | wc -l
