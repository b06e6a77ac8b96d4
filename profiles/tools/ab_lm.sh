#!/bin/bash
# LM-fused search: factor-row cache on / off, host frame loop, same box
echo "cache on:  $(python profiles/tools/time_lm_search.py | tail -1)"
echo "cache off: $(PDT_LM_CACHE=0 python profiles/tools/time_lm_search.py | tail -1)"
echo "cache on:  $(python profiles/tools/time_lm_search.py | tail -1)"
echo "host loop: $(PDT_CTC_LM_SEARCH=0 python profiles/tools/time_lm_search.py | tail -1)"
