"""Import-compatible stand-in for the one pytorch3d entry point the reference uses."""
