#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/ab_cfg.sh "c5" default heavy64 heavy192 2>&1
bash tools/ab_cfg.sh "c4" default resw4 resw6 2>&1
