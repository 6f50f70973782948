// Platform prelude of the scene API (counterpart of the reference's System.h:1-34): the standard
// headers the scene code relies on, `using namespace std`, and the per-TU generator `gen`.
// Unlike the reference, <algorithm> and <cstring> are included, so the code builds on Linux, and
// the generator can be seeded (RT_SEED) for reproducible runs.
#pragma once
#include <algorithm>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <limits>
#include <mutex>
#include <queue>
#include <random>
#include <ratio>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#define MULTI_THREAD 1

static std::random_device rd;
static std::mt19937 gen(rd());

using namespace std;
