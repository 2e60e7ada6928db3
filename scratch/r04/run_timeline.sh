#!/bin/bash
# GPU box: the timeline library is built in the container (scratch/r04/lib_timeline.so travels with the snapshot)
python scratch/r04/timeline_cs.py
