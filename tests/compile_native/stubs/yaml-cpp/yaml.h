// STAND-IN, NOT yaml-cpp (see ../Eigen/Dense): YAML::Node::operator[] and as<T>() as the reference's constructors
// use them (include/ESKF_LIO/Registration.hpp:23-28, LocalMap.hpp:28-37, CloudPreprocessor.hpp:20-32).
#pragma once
#include <string>
#include <vector>
namespace YAML {
class Node {
 public:
  Node operator[](const std::string&) const { return Node(); }
  bool IsDefined() const { return false; }   // yaml-cpp: whether the key exists
  template <typename T>
  T as() const { return T(); }
};
}  // namespace YAML
