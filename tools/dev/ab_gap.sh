# the dead time behind the last forward sweep under runtime settings: bash tools/dev/ab_gap.sh "VAR=val" ...
for v in "X_NONE=1" "$@"; do
  echo "== $v"
  env $v timeout -k 10 200 python tools/dev/tools_loss_section.py 2>&1 | grep -E "encoder done|^loss section" | cut -c1-140
done
