#!/bin/bash
# graph replay: executor / stream-count sweep (one process per setting)
run() { echo "== $*"; env "$@" CHECK=${CHECK:-0} CYCLES=4 timeout -k 10 300 python -u tools/graph_proto.py 2>&1 | grep -E "graphs=|Error|error|bit-exact|stats|Traceback|rc=|differ|^G |^D " ; }
CHECK=1 run SEMPYR_GRAPH_EXEC=own || exit 1
run SEMPYR_GRAPH_EXEC=own SEMPYR_REPLAY_STREAMS=2 || exit 1
run SEMPYR_GRAPH_EXEC=own SEMPYR_REPLAY_STREAMS=6 || exit 1
run SEMPYR_GRAPH_EXEC=own SEMPYR_REPLAY_STREAMS=1 || exit 1
run SEMPYR_GRAPH_EXEC=hip || exit 1
