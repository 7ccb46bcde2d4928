// ORACLE — TEST INFRASTRUCTURE ONLY.  NOT ROS: just enough of its surface for the reference's node classes to compile where
// they lie and to be driven in ONE process by a test harness: a node handle with a string parameter store, publishers and
// subscribers over an in-process topic bus (publish() queues a shared copy of the message, spinOnce() delivers the queue in
// order to every callback registered for the topic), a time stamp.  Nothing here computes anything about point clouds or poses.
#pragma once
#include <cstdint>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <typeindex>
#include <utility>
#include <vector>
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

namespace bus {
struct Topic {
  std::type_index type = std::type_index(typeid(void));
  std::vector<std::function<void(const std::shared_ptr<const void>&)>> callbacks;
};
struct State {
  std::map<std::string, Topic> topics;
  std::deque<std::pair<std::string, std::shared_ptr<const void>>> queue;
  std::map<std::string, std::string> params;      // parameter server of the private node handles: name -> text
};
inline State& state() { static State s; return s; }
inline void reset() { state() = State(); }
}  // namespace bus

inline bool ok() { return false; }          // the nodes' own spin loops end at once; the harness schedules them
inline void spinOnce() {                    // deliver everything queued, including what the callbacks publish meanwhile
  bus::State& s = bus::state();
  while (!s.queue.empty()) {
    auto m = s.queue.front();
    s.queue.pop_front();
    auto it = s.topics.find(m.first);
    if (it == s.topics.end()) continue;
    for (size_t k = 0; k < it->second.callbacks.size(); k++) it->second.callbacks[k](m.second);
  }
}

struct Publisher {
  std::string topic;
  template <class M> void publish(const M& msg) const {
    if (topic.empty()) return;
    bus::state().queue.emplace_back(topic, std::static_pointer_cast<const void>(std::make_shared<const M>(msg)));
  }
};
struct Subscriber {};
struct NodeHandle {
  bool getParam(const std::string& name, std::string& out) const {
    auto it = bus::state().params.find(name);
    if (it == bus::state().params.end()) return false;
    out = it->second;
    return true;
  }
  bool getParam(const std::string& name, float& out) const { std::string t; if (!getParam(name, t)) return false; out = std::stof(t); return true; }
  bool getParam(const std::string& name, double& out) const { std::string t; if (!getParam(name, t)) return false; out = std::stod(t); return true; }
  bool getParam(const std::string& name, int& out) const { std::string t; if (!getParam(name, t)) return false; out = std::stoi(t); return true; }
  bool hasParam(const std::string& name) const { return bus::state().params.count(name) != 0; }
  template <class M, class T> Subscriber subscribe(const std::string& topic, uint32_t, void (T::*fn)(const boost::shared_ptr<M const>&), T* obj) {
    bus::Topic& t = bus::state().topics[topic];
    t.type = std::type_index(typeid(M));
    t.callbacks.push_back([fn, obj](const std::shared_ptr<const void>& m) { (obj->*fn)(std::static_pointer_cast<const M>(m)); });
    return Subscriber();
  }
  template <class M> Publisher advertise(const std::string& topic, uint32_t) { return Publisher{topic}; }
};
}  // namespace ros
