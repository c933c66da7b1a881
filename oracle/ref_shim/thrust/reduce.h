#pragma once
#include "execution_policy.h"
namespace thrust {
template <typename In, typename T>
inline T reduce(device_t, In first, In last, T init) {
    for (; first != last; ++first) init = init + *first;
    return init;
}
}
