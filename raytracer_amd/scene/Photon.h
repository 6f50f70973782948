// Photon record of the scene API (reference: Photon.h:7-28).
#pragma once
#include "Color.h"
#include "Vect.h"

class Photon {
public:
    Vect position;
    Color power;
    char phi, theta;
    Vect dir;

    Photon() : position(0, 0, 0), power(0, 0, 0, 0), phi(0), theta(0) {}
    Photon(Vect pPos, Color pPower, char pPhi, char pTheta, Vect pDir) : position(pPos), power(pPower), phi(pPhi), theta(pTheta), dir(pDir) {}
};
