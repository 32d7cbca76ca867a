#!/bin/bash
# round 4: the evidence set on the final sources (PMC first: the bench quotes its traffic figure from it)
export LD_LIBRARY_PATH=$PWD/wxfactory_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
bash tools/collect_profiles.sh ${WX_TAG:-r04_v2} 2>&1 | tail -n 40
