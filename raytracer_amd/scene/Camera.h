// Camera: position + orthonormal basis holder (reference: Camera.h:6-30).
#pragma once
#include "../../include/rt_amd.h"
#include "Vect.h"

class Camera {
    Vect camPos, camDir, camRight, camDown;

public:
    Camera() : camPos(0, 0, 0), camDir(0, 0, 1), camRight(0, 0, 0), camDown(0, 0, 0) {}
    Camera(Vect pos, Vect dir, Vect right, Vect down) : camPos(pos), camDir(dir), camRight(right), camDown(down) {}
    Vect getCameraPosition() const { return camPos; }
    Vect getCameraDirection() const { return camDir; }
    Vect getCameraRight() const { return camRight; }
    Vect getCameraDown() const { return camDown; }

    rt_camera abi() const
    {
        rt_camera c;
        for (int i = 0; i < 3; i++) { c.pos[i] = camPos[i]; c.dir[i] = camDir[i]; c.right[i] = camRight[i]; c.down[i] = camDown[i]; }
        return c;
    }
};
