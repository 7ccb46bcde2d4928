// ORACLE — TEST INFRASTRUCTURE ONLY.  NOT boost: the name nanoflann_pcl.h asks for, bound to the standard shared pointer.
#pragma once
#include <memory>
namespace boost {
template <class T> using shared_ptr = std::shared_ptr<T>;
}
