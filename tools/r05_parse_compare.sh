#!/bin/bash
# parser CPU time per frame on the GPU box's cores: tools/bin/parse_prof_* (static builds of tools/dbg/parse_prof.cpp from two source states, made in the
# build container) on the committed 4K GOP and on its key frame alone
cd "$(dirname "$0")/.."
for b in ${BINS:-parse_prof_base parse_prof_new}; do
  for s in key_only parse_prof; do
    for i in 1 2; do echo -n "$b $s: "; EA1_PROF_INTER_ONLY=1 tools/bin/$b tools/bin/$s.stream 20 2>&1 >/dev/null | cut -c1-100 | tr '\n' ' '; echo; done
  done
done
