/* shim: see ../cub.cuh */
#pragma once
#include "../cub.cuh"
