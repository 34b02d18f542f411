# usage: tools/ab_probe.sh "<floor_probe args>" name1 name2 ...   (libraries ab_libs/<name>.so, same box, two passes)
args="$1"; shift
for rep in 1 2; do
for n in "$@"; do
echo "== $n (rep $rep)"
FSPT_LIB=$PWD/ab_libs/$n.so timeout 300 python3 tools/floor_probe.py 1920 1080 $args 2>&1 | grep wall
done; done
