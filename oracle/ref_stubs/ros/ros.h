// ORACLE — TEST INFRASTRUCTURE ONLY.  NOT ROS: inert declarations (a node handle without parameters, publishers that drop
// their message, a time stamp) so that the reference's MultiScanRegistration.cpp compiles where it lies.  Nothing here computes.
#pragma once
#include <cstdint>
#include <string>
#include <boost/shared_ptr.hpp>

#define ROS_INFO(...) ((void)0)
#define ROS_ERROR(...) ((void)0)
#define ROS_WARN(...) ((void)0)

namespace ros {
struct Duration {
  double s = 0;
  double toSec() const { return s; }
};
struct Time {
  uint32_t sec = 0, nsec = 0;
  Time& fromNSec(uint64_t t) { sec = (uint32_t)(t / 1000000000ull); nsec = (uint32_t)(t % 1000000000ull); return *this; }
  double toSec() const { return (double)sec + 1e-9 * (double)nsec; }
  Duration operator-(const Time& o) const { return Duration{((double)sec - (double)o.sec) + 1e-9 * ((double)nsec - (double)o.nsec)}; }
};
struct Rate {
  explicit Rate(double) {}
  bool sleep() { return true; }
};
inline bool ok() { return false; }          // the nodes' spin loops end at once
inline void spinOnce() {}
struct Publisher { template <class M> void publish(const M&) const {} };
struct Subscriber {};
struct NodeHandle {
  template <class T> bool getParam(const std::string&, T&) const { return false; }
  bool hasParam(const std::string&) const { return false; }
  template <class M, class T> Subscriber subscribe(const std::string&, uint32_t, void (T::*)(const boost::shared_ptr<M const>&), T*) { return Subscriber(); }
  template <class M> Publisher advertise(const std::string&, uint32_t) { return Publisher(); }
};
}  // namespace ros
