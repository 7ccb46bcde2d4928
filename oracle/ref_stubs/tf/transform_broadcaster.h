// ORACLE — TEST INFRASTRUCTURE ONLY.  NOT tf: a broadcaster that drops its transform.
#pragma once
#include <tf/transform_datatypes.h>
namespace tf {
struct TransformBroadcaster { void sendTransform(const StampedTransform&) {} };
}  // namespace tf
