#!/bin/bash
# Builds the emulation backend with AddressSanitizer + UBSan and runs the kernel bodies under it:
#   encoder_check.py  a few inputs through the whole encode pipeline in both parse modes (wave kernels on the SIMT emulator), vs the oracle
#   decoder_fuzz.py   300 corrupted member containers through the device decoder's kernel body
# Not part of the pytest tiers (slow, needs libasan); run by hand:  bash tests/sanitize/run.sh
set -e
cd "$(dirname "$0")/../.."
export ORZ_EMU_ASAN=/tmp/libemu_asan.so
g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -Wno-unknown-pragmas -shared -fPIC \
    -o $ORZ_EMU_ASAN tests/emu/emu_backend.cpp
make -s -C oracle all
export LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0
python tests/sanitize/encoder_check.py
python tests/sanitize/decoder_fuzz.py 1 300
