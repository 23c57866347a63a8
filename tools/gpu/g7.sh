cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/g7; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
python tools/step_breakdown.py > $O/breakdown.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
tail -3 $O/pytest.log; cut -c1-600 $O/bench.json
