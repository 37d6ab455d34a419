# development aid: k_tile time for (P2, P4) unroll factors (builds private copies of the library on the GPU box)
R=$GRAFT_REPO_ROOT
cd $R
for U in "3 3" "3 2" "3 4" "2 3" "4 3" "2 2"; do
  set -- $U
  sed -i "s/constexpr int T_UNROLL = [0-9];/constexpr int T_UNROLL = $1;/; s/constexpr int T_UNROLL4 = [0-9];/constexpr int T_UNROLL4 = $2;/" amplipy_amd/csrc/amp_tile.hpp
  python3 -c "import sys; sys.path.insert(0,'.'); from amplipy_amd import build; build.build(force=True)" > /dev/null 2>&1
  python3 bench.py --steps 12 --warmup 3 --cpu-passes 0 --no-pipeline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('unroll P2=$1 P4=$2 kernel_ms', d['roofline']['kernel_ms'], 'step', d['ms_per_step'])"
done
