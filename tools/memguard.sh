#!/bin/bash
# memguard.sh LIMIT_GB LOG -- command...   Run a command in its own session and end it (its process group, nothing
# else) if the resident memory of that session exceeds LIMIT_GB; the heaviest processes are written to LOG first.
# A GPU box ends the whole lease on a host-memory overrun, so risky steps run under this guard.
LIMIT_GB=$1; LOG=$2; shift 2; [ "$1" = "--" ] && shift
setsid "$@" &
PID=$!
while kill -0 $PID 2>/dev/null; do
    KB=$(ps -o rss= -s $PID 2>/dev/null | awk '{s+=$1} END {print s+0}')
    if [ "$KB" -gt $((LIMIT_GB * 1024 * 1024)) ]; then
        { echo "memguard: session $PID holds $((KB / 1024)) MiB > ${LIMIT_GB} GiB: ending it"; ps -o pid,rss,nlwp,args -s $PID --sort=-rss | head -15; } >> "$LOG"
        kill -TERM -- -$PID 2>/dev/null; sleep 2; kill -KILL -- -$PID 2>/dev/null
        wait $PID 2>/dev/null
        exit 99
    fi
    sleep 0.5
done
wait $PID
