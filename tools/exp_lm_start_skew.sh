for sk in 0 300 600 900 1200 2000; do for sh in 8 0 4; do
  echo "skew=$sk shift=$sh $(IDSP_DIAG=1 IDSP_LM_SKEW=$sk IDSP_LM_SKEW_SHIFT=$sh python bench.py --config c2 --layout lane --no-cpu --no-c5 --no-c3 --no-c4 --no-lane-major --no-inplace --steps 50 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['frac'], d['integrity']['match'])")"
  [ $sk = 0 ] && break
done; done
