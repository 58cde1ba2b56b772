#!/bin/bash
# round 6: dispatch rules re-swept for the two-piece forms (weight-gradient workgroup target, XCD map threshold), step level, one box
cd "$(dirname "$0")/.."
for p in fp32x3b2 fp32x3h2 fp32x3; do
  echo "== $p"
  bash tools/ab_env.sh "--precision $p" - OSVOS_W3_WANT=192 OSVOS_W3_WANT=320 OSVOS_W3_WANT=384 OSVOS_X3_MAP_FACTOR=9 OSVOS_X3_MAP_FACTOR=81
done
