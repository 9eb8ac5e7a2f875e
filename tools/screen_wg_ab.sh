#!/bin/bash
# screen_wg_ab.sh: config 3's greedy generation (tools/gen_graph_probe.py: ms per batch, eager and as a replayed hipGraph) with ONE pipelined
# screening workgroup per CU (a variant build whose D = 64 candidate list is 48 KB again, planned for one: the state until round 6) against
# TWO (the product), same box, alternating.  (D = 128 - config 4 - was measured the same way while the product still ran two there:
# profiles/r06_screen_wg_ab.txt; it runs one.)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
mkdir -p build/variants
[ -f build/variants/screen_old_list.so ] || bash tools/build_variant_tu.sh catalog_bf16 build/variants/screen_old_list.so -DSCREEN_PIPE_ENTRIES_64=6144 > /dev/null 2>&1
for rep in 1 2; do
  echo "== config 3, one workgroup per CU (48 KB list)"; PCVAE_LIB=$ROOT/build/variants/screen_old_list.so PCVAE_SCREEN_WG_PER_CU=1 python3 tools/gen_graph_probe.py 3 2>/dev/null | head -1
  echo "== config 3, two workgroups per CU (product)"; python3 tools/gen_graph_probe.py 3 2>/dev/null | head -1
done
