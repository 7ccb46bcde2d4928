// The C-ABI handles of the three single-sweep nodes, shared by the translation units that link them (loamx_odom_process_linked,
// loamx_map_process_linked: a sweep's clouds handed from node to node in HBM instead of through host messages).
#pragma once
#include "features.hpp"
#include "odometry.hpp"

struct loamx_scanreg {
  loamx::FeatureExtractor fx;
  explicit loamx_scanreg(int device) : fx(device) {}
};

struct loamx_odom {
  loamx::OdometryBatch od;
  explicit loamx_odom(int device) : od(device, 1) {}
};
