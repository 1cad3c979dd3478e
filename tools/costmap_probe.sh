for opt in "sort.costmap=0" "sort.costmap=1" "sort.costmap=0 --option sort.reverse=1" "sort.costmap=1 --option sort.reverse=1"; do
  tag=$(echo "$opt" | tr -c 'a-z0-9=' '_')
  python bench.py --cpu-seconds 0 --option $opt > gpurun_out/b_cm_$tag.json 2> gpurun_out/b_cm_$tag.err || exit 1
done
