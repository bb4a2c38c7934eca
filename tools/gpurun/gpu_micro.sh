#!/bin/bash
timeout -k 10 120 ./tools/micro/valu_rate
