# both profile sessions (C2 + tests + configs, C3) in one call
bash tools/sessions/session_r04_prof.sh final
bash tools/sessions/session_r04_c3.sh
