#pragma once
#include <type_traits>
// the real macro registers a factory; what can be checked without ROS is that the class is a concrete Base
#define PLUGINLIB_EXPORT_CLASS(Class, Base)                                                             \
  static_assert(std::is_base_of<Base, Class>::value, #Class " must derive from " #Base);               \
  static_assert(!std::is_abstract<Class>::value, #Class " leaves a pure virtual of " #Base " open");
