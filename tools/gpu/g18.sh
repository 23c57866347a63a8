cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/g18; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
