set -x
O=gpurun_out/r03_a; mkdir -p $O
python -m pytest tests/test_gpu_scorer.py -m gpu -x -q -k "install_alone or overflow or config5 or plugin_surface or outlier" > $O/t1.log 2>&1; tail -5 $O/t1.log
python -m pytest tests/test_train_step.py -m gpu -x -q -k "gradient_only" > $O/t2.log 2>&1; tail -3 $O/t2.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json; tail -5 $O/bench.err
python bench.py --trace burst > $O/trace_burst.json 2> $O/trace_burst.err; cat $O/trace_burst.json; tail -3 $O/trace_burst.err
python bench.py --trace gamma > $O/trace_gamma.json 2> $O/trace_gamma.err; cat $O/trace_gamma.json; tail -3 $O/trace_gamma.err
python -m pytest tests/test_gpu_distributed.py -m gpu -x -q -s > $O/t3.log 2>&1; tail -8 $O/t3.log
