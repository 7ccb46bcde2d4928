#!/bin/sh
# The ONE change a maintainer makes to a wrapper source: in the reference's own
# src/lib/MultiScanRegistration.cpp, the per-point host loop of MultiScanRegistration::process (the whole function body)
# becomes one call into the device path — ring binning, relative times and the IMU de-skew run in
# loamx_scanreg_process_raw behind the adapter's processRawSweep().  Everything else in the file (MultiScanMapper, parameter
# handling, the start-up delay, the subscriber) stays the reference's code, byte for byte.
#
#   usage: patch_multiscan_registration.sh <reference>/src/lib/MultiScanRegistration.cpp > MultiScanRegistration.cpp
#
# The product ships this recipe, not a copy of the file.
set -e
[ -r "$1" ] || { echo "usage: $0 <reference>/src/lib/MultiScanRegistration.cpp" >&2; exit 2; }
awk '
/^void MultiScanRegistration::process\(/ {
  print
  print "{"
  print "  // loamx: the sweep goes to the device as it came; binning, relative times and IMU de-skew happen there"
  print "  processRawSweep(scanTime, laserCloudIn, _scanMapper.getLowerBound(), _scanMapper.getUpperBound(), _scanMapper.getNumberOfScanRings());"
  print "  publishResult();"
  print "}"
  skipping = 1
  next
}
skipping && /^}/ { skipping = 0; next }
!skipping { print }
' "$1"
