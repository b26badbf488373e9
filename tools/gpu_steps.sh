#!/bin/bash
# Runs the given steps ("name|seconds|command" per argument) one after the other on
# the GPU box, each under its own timeout, logging to gpurun_out/<name>.log. An
# ordinary failure lets the next step run; a step that hits its timeout (or is
# killed) ends the whole call: nothing else touches the GPU after a hang.
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
for step in "$@"; do
    name="${step%%|*}"; rest="${step#*|}"; secs="${rest%%|*}"; cmd="${rest#*|}"
    echo "== $name (limit ${secs}s): $cmd"
    start=$(date +%s)
    timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2> "gpurun_out/$name.err"
    rc=$?
    echo "== $name rc=$rc in $(( $(date +%s) - start ))s"
    tail -n 5 "gpurun_out/$name.log"
    if [ $rc -ne 0 ]; then tail -n 15 "gpurun_out/$name.err"; fi
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "== $name hit its limit: stopping here"
        exit $rc
    fi
done
exit 0
