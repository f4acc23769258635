"""oracle/ -- CPU checkers for the GeoT segment-reduction hot path.

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this package.  ``geot_amd`` never does.

* ``oracle.api``  -- numpy/ctypes front end of ``libgeot_oracle.so`` (our C restatement,
  ``geot_oracle.c``; every function cites the reference file:line it follows).
* ``oracle.ref``  -- front end of ``oracle/_ref/libgeot_ref*.so``: the reference's OWN CPU
  ``index_scatter`` compiled in place from ``/root/reference`` (``make -C oracle ref``).
"""
