# final: tests, both profile sessions, then the three full bench lines
bash tools/sessions/session_r04_prof.sh final
bash tools/sessions/session_r04_c3.sh
