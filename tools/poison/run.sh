#!/bin/bash
# Runs a command with every hipMalloc poisoned (see poison_alloc.c).  Usage: tools/poison/run.sh <command ...>
here="$(cd "$(dirname "$0")" && pwd)"
gcc -O1 -shared -fPIC -I/opt/rocm/include "$here/poison_alloc.c" -o /tmp/libpoison_alloc.so -ldl -L/opt/rocm/lib -lamdhip64 || exit 1
LD_PRELOAD=/tmp/libpoison_alloc.so "$@"
