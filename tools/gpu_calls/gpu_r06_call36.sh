#!/bin/bash
# round 6, call 36: which counters exist for address translation / L2 / memory-side traffic
set -u
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "utcl|tlb|translat|TCC_EA|TCC_HIT|TCC_MISS|MALL|TCC_TAG_STALL|TCC_.*STALL|TCP_PENDING|TCP_TCC|TCC_REQ|TCC_BUBBLE|TCC_EA0_WR_UNCACHED|TCC_.*DRAM|TCC_.*CREDIT" | sed 's/^[ \t]*//' | cut -c1-200 | sort -u | head -120
