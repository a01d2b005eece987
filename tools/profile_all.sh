#!/bin/bash
# GPU box: the round's full-size profiles of every BASELINE configuration and mode (tools/profile_round.sh each), ~15 min.
TAG=${1:-r05}
STEPS=5 tools/profile_round.sh $TAG 3
STEPS=5 tools/profile_round.sh $TAG 3 "--fast"
STEPS=5 tools/profile_round.sh $TAG 2
STEPS=5 tools/profile_round.sh $TAG 5
STEPS=5 tools/profile_round.sh $TAG 5 "--fast"
STEPS=5 tools/profile_round.sh $TAG 6
STEPS=5 tools/profile_round.sh $TAG 6 "--fast"
STEPS=3 tools/profile_round.sh $TAG 4 "--cells 12500"
STEPS=3 tools/profile_round.sh $TAG 4 "--cells 12500 --fast"
