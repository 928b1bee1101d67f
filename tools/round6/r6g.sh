R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6g; mkdir -p $O
cd $R
python -c "
import sys; sys.path.insert(0,'.')
from mods_amd import synthetic
synthetic.blob_image(768,1024,4000,12345).astype('uint8').tofile('/tmp/img.u8')"
timeout 600 tools/ubench/mser_tree /tmp/img.u8 768 1024 64 256 864 2048 > $O/mser_tree.txt 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
( time python bench.py --config wxbs > $O/bench_wxbs.json 2> $O/bench_wxbs.err ) 2> $O/bench_wxbs.time
cat $O/mser_tree.txt; grep -n "passed\|failed" $O/pytest.txt; tail -3 $O/bench_default.time $O/bench_wxbs.time; tail -3 $O/bench_default.err $O/bench_wxbs.err
python tools/bench_line.py default < $O/bench_default.json; python tools/bench_line.py wxbs < $O/bench_wxbs.json
