for round in 1 2; do
for v in g_b16 g_b4 g_p4_16; do
  echo -n "$v: "; PCVAE_LIB=$PWD/build/variants/$v.so python tools/bench_gather.py 2>&1 | tail -1
done; done
for v in g_b16 g_b4 g_p4_16; do
  echo -n "x8 $v: "; PCVAE_LIB=$PWD/build/variants/$v.so python tools/bench_gather.py --mult 8 --tables 2 2>&1 | tail -1
done
