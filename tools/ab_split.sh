# same-box A/B (GPU box): compiler-generated f16 operand split (-DWN_SPLIT_PLAIN) against the hand-written one
for v in "plain:-DWN_SPLIT_PLAIN" "asm:" "plain2:-DWN_SPLIT_PLAIN" "asm2:"; do
  n=${v%%:*}; f=${v#*:}
  bash tools/variant.sh $n "$f" "grads_64"
done
