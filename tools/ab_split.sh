# same-box A/B: compiler-generated operand split against the hand-written one
for v in "plain:-DWN_SPLIT_PLAIN" "f16asm:-DWN_SPLIT_PLAIN_BF16" "asm:" "plain2:-DWN_SPLIT_PLAIN" "f16asm2:-DWN_SPLIT_PLAIN_BF16" "asm2:"; do
  n=${v%%:*}; f=${v#*:}
  bash tools/variant.sh $n "$f" "grads_64"
done
