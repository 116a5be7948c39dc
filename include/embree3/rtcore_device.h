/* Forwarder: the whole embree3 contract lives in rtcore.h (reference twin: include/embree3/rtcore_device.h). */
#pragma once
#include "rtcore.h"
