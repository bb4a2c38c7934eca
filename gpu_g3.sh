#!/bin/bash
timeout -k 10 300 python tools/hbm_probe.py
