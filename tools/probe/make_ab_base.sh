#!/bin/bash
# Builds the libraries of the last COMMIT into ab/base/ (the A arm of tools/probe/ab_bench.sh) and leaves the working tree's build
# in place: run before `gpurun -- bash tools/probe/ab_bench.sh` to A/B uncommitted kernel changes on one box.
set -e
cd "$(dirname "$0")/../.."
C=listening-to-sound-of-silence-for-speech-denoising_amd/csrc
git stash -q
trap 'git stash pop -q' EXIT
make -C $C -j8 > /dev/null
mkdir -p ab/base
cp listening-to-sound-of-silence-for-speech-denoising_amd/libsos_hip.so listening-to-sound-of-silence-for-speech-denoising_amd/libsos_hip_f16.so ab/base/
trap - EXIT
git stash pop -q
make -C $C -j8 > /dev/null
ls -la ab/base
