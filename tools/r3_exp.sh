for shape in 32768x3456x1024 32768x1024x1024 32768x1024x512; do
  for sp in 0 3 4 5 6 7 9 12 16 24 32; do
    echo -n "$shape split $sp: "; FFH_BF16_DMA_SPLIT=$sp timeout 120 python tools/bf16_twin_probe.py $shape 2>/dev/null | grep "operand twins" | sed 's/.*dW/dW/'
  done
done
