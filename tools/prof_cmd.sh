#!/bin/bash
# rocprofv3 kernel-trace summary of any command (through gpurun).  $1 = tag, $2 = kernel-name filter for the echo, rest = command
R=$GRAFT_REPO_ROOT
tag=$1; filt=$2; shift; shift
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rpc_$tag
rocprofv3 --kernel-trace --stats -d /tmp/rpc_$tag -o p -- "$@" > /tmp/rpc_$tag.log 2>&1
DB=$(find /tmp/rpc_$tag -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $DB $R/gpurun_out/prof/cmd_$tag.txt "$*" > /dev/null
echo "== $tag: $(grep -v rocprofv3 /tmp/rpc_$tag.log | tail -1 | cut -c1-100)"
grep -h "$filt" $R/gpurun_out/prof/cmd_$tag.txt
[ -n "$TIMELINE" ] && python $R/tools/rocprof_timeline.py $DB $TIMELINE > $R/gpurun_out/prof/timeline_$tag.txt
