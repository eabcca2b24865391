#!/bin/bash
export TMPDIR=/tmp
echo "== host profile: dqn_pixel_per_device"; timeout 200 python tools/prof_agents.py dqn_pixel_per_device 3000 2>/dev/null | cut -c1-160 | head -45
echo "== host profile: c51_pixel_uniform_device"; timeout 200 python tools/prof_agents.py c51_pixel_uniform_device 3000 2>/dev/null | cut -c1-160 | head -30
