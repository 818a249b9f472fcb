"""Import path only.  The reference's matplotlib debug viewers (`algorithm/utils/visualization/`) are control-plane
UI and out of scope here; three of its plugin files import the two class names at module
level, so inert stand-ins keep those files loadable."""
