"""MI355X-native ORB front-end (FAST -> orientation -> steered BRIEF -> Hamming top-2)."""
