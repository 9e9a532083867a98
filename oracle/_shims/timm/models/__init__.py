"""Import shim (test tooling only): lets /root/reference import without timm.
Written from the reference's call sites; provides only the three names it uses."""
