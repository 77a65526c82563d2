# window-attention kernel on the Swin shapes (stage-3 of Swin-B/384 at batch 128, a shifted block, the 224-pixel models): tools/wattn_shapes.sh
cd $GRAFT_REPO_ROOT
for shape in "512 16 144 0" "2048 8 144 64" "2048 16 49 0" "8192 8 49 64"; do
  set -- $shape
  timeout -k 10 120 python tools/wattn_bench.py --windows $1 --heads $2 --n $3 --mask $4 --rel 2>/dev/null | tail -1
done
