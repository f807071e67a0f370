for opt in "" "knn_sample_tiles=1536" "knn_sample_tiles=1024" "knn_sample_tiles=1536,knn_sample_every=64" "knn_sample_every=16" "knn_sample_tiles=3600"; do
  echo "== c4 $opt"; DDX_OPTIONS="$opt" python profiles/tools/knn_cells_check.py 500000 33000 0.02 0 500 2>&1 | grep "^cells.*wall\|different"
done
for opt in "" "knn_sample_tiles=384" "knn_sample_tiles=768" "knn_sample_every=16" "knn_sample_every=64"; do
  echo "== headline $opt"; DDX_OPTIONS="$opt" python profiles/tools/knn_cells_check.py 100000 30000 0.03 0 2000 2>&1 | grep "^cells.*wall\|different"
done
