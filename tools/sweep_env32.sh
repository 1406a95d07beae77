set -e
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
NAME=$1; shift
for v in "$@"; do
  echo "== fp32 $NAME=$v"
  env $NAME=$v timeout -k 10 120 python3 tools/sweep_wave.py obj --mode fp32 --spp 32 --width 3840 --height 2160 --count 2>&1 | grep "^obj" | sed 's/| lane-eff.*| per ray/| per ray/'
done
